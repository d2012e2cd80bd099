"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into the per-kernel stats table that
`rocprofv3 --stats` prints: calls, total/avg/min/max duration (ns), share.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                      "max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, n, s, avg, mn, mx, vg, ag, lds, gx, wx in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{short}` | {n} | {s/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.2f} | {vg} | {ag} | {lds} | {gx} | {wx} |")
    print(f"\ntotal kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
