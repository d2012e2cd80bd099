"""CPU-only checks of the host side: the C-ABI library builds/loads and exports every declared
symbol (no compute calls), the drop-in module mirrors the reference's parameter tree, and the RoPE
table matches the oracle's complex table."""
import ctypes
import os

import pytest
import torch

from oracle import dit_oracle as O


def test_library_builds_and_exports_header_symbols():
    """The dynamic symbol table of the product library IS include/chronoedit_hip.h - no more (cross-TU launch helpers, kernel stubs and
    the kernel-body selectors are not ABI), no less; the diagnostic build adds exactly include/chronoedit_hip_diag.h."""
    from chronoedit_amd import hiplib
    path = hiplib.build()
    assert os.path.exists(path) and os.path.exists(hiplib.DIAG_LIB_PATH)
    lib = ctypes.CDLL(path)
    syms = hiplib.header_symbols()
    assert len(syms) >= 50
    for s in syms:
        assert hasattr(lib, s), s
    assert set(hiplib.SIGNATURES) == set(syms)
    assert set(hiplib.exported_symbols(path)) == set(syms)                      # `nm -D --defined-only` == the header
    diag = hiplib.header_symbols(diag=True)
    assert set(diag) == set(hiplib.DIAG_SIGNATURES) and all(d.startswith(("ce_set_", "ce_diag_")) for d in diag)
    assert set(hiplib.exported_symbols(hiplib.DIAG_LIB_PATH)) == set(syms) | set(diag)
    for d in diag:                                                              # two engines in one process cannot fight over a selector:
        assert not hasattr(lib, d), d                                           # the product library has none
    lib.ce_build_info.restype = ctypes.c_int
    assert lib.ce_build_info() == 0x10                                          # product build: no diagnostics, F8_DMA_SCHED 1, no ablation
    dl = ctypes.CDLL(hiplib.DIAG_LIB_PATH)
    dl.ce_build_info.restype = ctypes.c_int
    assert dl.ce_build_info() == 0x11


def test_loader_refuses_a_diagnostic_build_as_the_product_library(monkeypatch):
    """CE_HIPLIB_PATH only redirects load() (with a warning), and a build that says of itself that it is not the product is refused."""
    from chronoedit_amd import hiplib
    hiplib.build()
    monkeypatch.setattr(hiplib, "_LIB", None)
    monkeypatch.setenv("CE_HIPLIB_PATH", hiplib.DIAG_LIB_PATH)
    with pytest.warns(RuntimeWarning, match="CE_HIPLIB_PATH"):
        with pytest.raises(RuntimeError, match="diagnostic build"):
            hiplib.load()
    monkeypatch.setenv("CE_HIPLIB_ALLOW_DIAGNOSTIC_BUILD", "1")
    with pytest.warns(RuntimeWarning):
        assert hiplib.load().ce_build_info() & 1
    monkeypatch.setattr(hiplib, "_LIB", None)


def test_param_tree_matches_reference_names():
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=96, image_dim=64, added_kv_proj_dim=256)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=96,
                                     image_dim=64, added_kv_proj_dim=256, device="cpu")
    own = {k: tuple(v.shape) for k, v in m.named_parameters()}
    ref = O.param_shapes(cfg)
    assert own == ref
    for k, v in m.named_parameters():
        assert (v.dtype == torch.float32) == O.keep_fp32(k), k
    assert m.config.patch_size == (1, 2, 2) and m.dtype == torch.bfloat16


def test_golden_reference_state_dict_names(golden_dir):
    """The synthetic dict is accepted by the reference's own module (checked when the fixtures were made);
    here: the 14B tree has the expected parameter count (16.395 B)."""
    n = sum(int(torch.tensor(s).prod()) for s in O.param_shapes(O.DiTConfig()).values())
    assert abs(n / 1e9 - 16.395) < 0.01


@pytest.mark.parametrize("T", [2, 8])
def test_rope_table_matches_oracle(T):
    from chronoedit_amd.transformer import rope_cos_sin
    cfg = O.DiTConfig(num_attention_heads=2, num_layers=1)
    cs = rope_cos_sin(128, 1024, 8, T, 6, 10)
    ref = O.rope_table(cfg, T, 12, 20)[0, 0]
    assert cs.shape == (T * 60, 64, 2)
    assert torch.allclose(cs[..., 0].double(), ref.real, atol=1e-7) and torch.allclose(cs[..., 1].double(), ref.imag, atol=1e-7)
    with pytest.raises(AssertionError):
        rope_cos_sin(128, 1024, 8, 5, 6, 10)


def test_cpu_call_fails_loudly():
    from chronoedit_amd import ops
    with pytest.raises(ops.HipKernelError):
        ops.ln_affine(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64), torch.zeros(64), 1e-6)


def test_product_side_shape_helpers_match_the_oracles():
    """bench.py / tools use product-side helpers (they must not import oracle/ outside the cpu_baseline leg):
    the flop count and the VAE parameter table agree with the oracle's."""
    from chronoedit_amd.flops import dit_flops_per_forward
    from chronoedit_amd.vae import wan_vae_param_shapes
    from oracle import dit_oracle as O
    from oracle import vae_oracle as V
    for n in (512, 7200, 13068, 28800):
        assert dit_flops_per_forward(n) == float(O.flops_per_forward(O.DiTConfig(), n))
    assert abs(dit_flops_per_forward(7200) / 1e12 - 222.38) < 0.01
    assert dit_flops_per_forward(7200, num_layers=2) == float(O.flops_per_forward(O.DiTConfig(num_layers=2), 7200))
    a, b = wan_vae_param_shapes(), V.param_shapes(V.VAEConfig())
    assert set(a) == set(b) and all(tuple(a[k]) == tuple(b[k]) for k in a)


def test_scaling_model_runs_on_the_committed_one_gpu_line():
    """tools/scaling_model.py prices the sharded step from profiles/r02_bench_n28800_one_gpu.json: the
    prediction must be reproducible from the repository alone, every split must beat one GPU, and the defaults bench.py picks per
    GPU count (guidance-pair split on 2, one Ulysses group from 4 on) must be the model's best or within 5 % of it."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("scaling_model", os.path.join(root, "tools", "scaling_model.py"))
    sm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sm)
    one = json.load(open(os.path.join(root, "profiles", "r02_bench_n28800_one_gpu.json")))
    base = one["ms_per_step"]
    best = {}
    for W in (2, 4, 8):
        r = {"seq": sm.model(W, False, 55.0, 25.0, one), "batched": sm.model(W, False, 55.0, 25.0, one, pair_batched=True),
             "cfgp": sm.model(W, True, 55.0, 25.0, one)}
        for v in r.values():
            assert v["step_ms"] < base and 0.5 < base / v["step_ms"] / W <= 1.0, v
        best[W] = r
    # 2 GPUs: split the guidance pair (one xGMI link between two GPUs); 4, 8: one Ulysses group with the pair batched is the best split
    assert best[2]["cfgp"]["step_ms"] < min(best[2]["seq"]["step_ms"], best[2]["batched"]["step_ms"])
    for W in (4, 8):
        assert best[W]["batched"]["step_ms"] <= min(best[W]["seq"]["step_ms"], best[W]["cfgp"]["step_ms"])
    r8 = best[8]["batched"]
    assert r8["rows"] == 2 * 3648 and r8["heads"] == 5  # shards rounded up to 64 tokens, two samples stacked


def test_vt_column_padding_helper():
    """ops.vt_columns: whole 64-key strips plus one spare strip (the last tile of the last sample reads a whole strip), rows 16-B aligned."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "chronoedit_amd", "ops.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "vt_columns")
    ns = {}
    exec(compile(ast.Module([fn], []), "ops.vt_columns", "exec"), ns)
    for n in (1, 63, 64, 65, 7200, 14400, 26136, 28800):
        c = ns["vt_columns"](n)
        assert c % 64 == 0 and c >= (n + 63) // 64 * 64 + 64 - 64 and c - n >= 64 and c % 8 == 0, (n, c)


def test_blocked_layout_tile_to_block_magic_is_exact():
    """ce_attention_vt_blocked_bf16 finds the source block of key tile t with one multiply-high: (t * ceil(2^32 / tps)) >> 32 must equal
    t // tps for every tile index a sequence can have (tiles of 64 keys: 2^16 tiles = 4 M keys) and every block size in tiles."""
    import numpy as np
    t = np.arange(1 << 16, dtype=np.uint64)
    for tps in list(range(2, 130)) + [225, 226, 450, 451, 900, 1024, 4095, 65535]:
        magic = np.uint64(((1 << 32) + tps - 1) // tps)
        assert magic < (1 << 32)
        assert np.array_equal((t * magic) >> np.uint64(32), t // np.uint64(tps)), tps


@pytest.mark.parametrize("KT,T_out,H,W,Cin,Cout", [(3, 2, 5, 6, 64, 16), (1, 3, 4, 7, 128, 8), (3, 1, 3, 3, 192, 8), (3, 2, 4, 5, 96, 8),
                                                   (1, 1, 3, 4, 96, 8)])
def test_conv3d_gemm_address_map_reproduces_the_convolution(KT, T_out, H, W, Cin, Cout):
    """ce_conv3d_gemm_bf16 hands a stride-1 3x3(x3) convolution to the GEMM as: A row r = the input stack read linearly from position r
    (lda = Cin), K walked in 64-wide tiles whose source offset is t*64 + (t // tiles(S)) * extra1 + (t // tiles(3 S)) * extra2 elements
    (S = one (kt, kh) run = 3 Cin rounded up to whole tiles: Cin = 96 reads 32 channels of the next pixel against zero weights), C row
    r = padded output position r + Wp + 1, an odd tile count padded by one tile of zero weights, borders zeroed afterwards.  This
    replays exactly that arithmetic on the CPU (gather + matmul) against torch's conv3d."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(KT * 100 + Cin)
    Hp, Wp = H + 2, W + 2
    n_in = T_out + KT - 1
    stack = torch.zeros(n_in + 1, Hp, Wp, Cin, dtype=torch.float64)  # + one zeroed slack frame
    stack[:n_in, 1:-1, 1:-1] = torch.randn(n_in, H, W, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, KT, 3, 3, generator=g, dtype=torch.float64).to(torch.bfloat16).double()
    # what conv3d computes on those frames (zero spatial padding = the border; the front frames are IN the stack already)
    x = stack[:n_in, 1:-1, 1:-1].permute(3, 0, 1, 2)[None]  # [1, Cin, n_in, H, W]
    want = F.conv3d(x, w, padding=(0, 1, 1))[0].permute(1, 2, 3, 0)  # [T_out, H, W, Cout]

    seg1 = (3 * Cin + 63) // 64 * 64
    k = KT * 3 * seg1
    kpad = ((k // 64) + 1) // 2 * 2 * 64
    from chronoedit_amd.vae import _ConvPack
    wg = _ConvPack(w.to(torch.bfloat16), None).gemm_weight().double()  # the host-side packing the engine hands to the kernel
    assert wg.shape == (Cout, kpad)
    stride1, seg2, stride2 = Wp * Cin, 3 * seg1, Hp * Wp * Cin
    t = torch.arange(kpad // 64)
    tile_off = t * 64 + (t // (seg1 // 64)) * (stride1 - seg1) + (t // (seg2 // 64)) * (stride2 - (seg2 // seg1) * stride1)
    koff = (tile_off[:, None] + torch.arange(64)[None]).reshape(-1)  # element offset of GEMM column k from the row's base
    rows = T_out * Hp * Wp - 2 * (Wp + 1)
    flat = stack.reshape(-1)
    idx = torch.arange(rows)[:, None] * Cin + koff[None]
    assert int(idx.max()) < flat.numel()  # the padded tile stays inside the slack frame
    c = flat[idx] @ wg.t()  # [rows, Cout]
    out = torch.zeros(T_out * Hp * Wp, Cout, dtype=torch.float64)
    out[Wp + 1 : Wp + 1 + rows] = c
    out = out.reshape(T_out, Hp, Wp, Cout)
    assert torch.allclose(out[:, 1:-1, 1:-1], want, atol=1e-9)


def test_macro_tile_choice_of_the_step_shapes():
    """ce_gemm_bf16_tile_rows: the automatic 384 x 256 / 288 x 256 / 256 x 256 macro-tile choice is a pure function of the shape, the CU count and
    the split-K workspace - pinned here for the five large GEMMs of a block at BOTH row counts of the 720p step (M = 14 400: guidance pair
    batched, BASELINE configs[1]; M = 7 200: the distilled B = 1 step, configs[2]) on 256 CUs with the engine's 96 MiB workspace: what the
    same-box A/B measured as the faster tile for each (profiles/r06_gemm_tile_choice.txt; round 6 refit)."""
    from chronoedit_amd import hiplib, ops
    lib = hiplib.load()
    ws = ops.GEMM_WS_BYTES
    assert ws == 256 * 384 * 256 * 4  # one slab of the largest macro tile per CU: every split the dispatcher may choose fits
    pick = lambda M, N, K: lib.ce_gemm_bf16_tile_rows(M, N, K, 256, ws)
    assert pick(14400, 10240, 5120) == 384   # q | k (level)
    assert pick(5120, 14400, 5120) == 256    # V^T: measured 256 by 10 %
    assert pick(14400, 5120, 5120) == 384    # out-projections, q of the cross-attention: 384 by 6 %
    assert pick(14400, 13824, 5120) == 256   # FFN-up: 256 by 2.3 %
    assert pick(14400, 5120, 13824) == 384   # FFN-down: 384 by 3.6 %
    assert pick(7200, 10240, 5120) == 384    # q | k: 384 by 11.5 %
    assert pick(5120, 7200, 5120) == 256     # V^T: 256 by 4.5 %
    assert pick(7200, 5120, 5120) == 288     # out-projections, q of the cross-attention: the 288-row form (25 whole row tiles, 1.95 rounds, no slabs) by 17 %
    assert pick(7200, 13824, 5120) == 384    # FFN-up: 384 by 3.0 % (the round-3 model said 256)
    assert pick(7200, 5120, 13824) == 384    # FFN-down: 384 by 2.8 % - needs the 96 MiB scratch: 124 tail tiles x 2 slabs of 384 rows
    # the 288-row form nowhere else on the engine's shapes (its loop is 2-6 % behind the 384-row one; measured slower wherever its tail needs slabs)
    for M in (14400, 13068, 26136, 28800, 57600, 3648, 7296):
        for N, K in ((10240, 5120), (5120, 5120), (13824, 5120), (5120, 13824)):
            assert pick(M, N, K) != 288, (M, N, K)
        assert pick(5120, (M + 63) // 64 * 64, 5120) != 288
    assert pick(5120, 7232, 5120) == 256
    assert pick(0, 5120, 5120) == 0 and pick(256, 256, 32) == 0
    # without a workspace a partial last round cannot be cut along K
    assert lib.ce_gemm_bf16_tile_rows(14400, 5120, 5120, 256, 0) == 384
    assert lib.ce_gemm_bf16_tile_rows(14400, 13824, 5120, 256, 0) == 256
    assert lib.ce_gemm_bf16_tile_rows(7200, 5120, 13824, 256, 64 << 20) != 384  # the 64 MiB scratch of rounds 1-5: 384 cannot cut its tail (-13 %)


def test_mx_scale_layout_helpers_agree_with_the_kernels_offset_formula():
    """The tiled E8M0 scale layout of the MX fp8 GEMM operands ([rows / 128][K / 128][4 blocks][16 rows][8 row groups], ce_common.h
    `mx_gemm_scale_offset`): `ops.mx_scale_bytes` sizes it and `ops.mx_scales_to_rows` undoes it - replayed here against the device
    formula ((row >> 7) ktiles + (blk >> 2)) 512 + (blk & 3) 128 + (row & 15) 8 + ((row >> 4) & 7), incl. a ragged last row tile."""
    from chronoedit_amd import ops
    for rows, K in ((128, 128), (300, 512), (1003, 5120), (17, 256)):
        ktiles = K // 128
        buf = torch.zeros(ops.mx_scale_bytes(rows, K), dtype=torch.uint8)
        assert buf.numel() == (rows + 127) // 128 * ktiles * 512
        want = torch.randint(1, 255, (rows, K // 32), dtype=torch.uint8, generator=torch.Generator().manual_seed(rows))
        r = torch.arange(rows)[:, None]
        b = torch.arange(K // 32)[None, :]
        off = ((r >> 7) * ktiles + (b >> 2)) * 512 + (b & 3) * 128 + (r & 15) * 8 + ((r >> 4) & 7)
        assert int(off.max()) < buf.numel() and off.unique().numel() == off.numel()  # a bijection onto distinct bytes
        buf[off.reshape(-1)] = want.reshape(-1)
        assert torch.equal(ops.mx_scales_to_rows(buf, rows, K), want)


@pytest.mark.parametrize("KT,T,H,W,Cout", [(3, 2, 11, 21, 3), (1, 1, 8, 16, 4)])
def test_head_conv_row_packing_reproduces_the_convolution(KT, T, H, W, Cout):
    """The arithmetic of conv_head_kernel (csrc/ce_conv.hip) replayed on the CPU: the three kernel ROWS ride in the matrix instruction's
    output rows - A row n = (kh, co), B = 16 consecutive pixels of ONE padded input row rho - so P[rho][(kh, co)][pixel], summed over
    (kt, kw, 32-channel chunk), holds what input row rho contributes to the output rows rho - kh, and out[h] = P[h][0] + P[h+1][1] +
    P[h+2][2].  A wave owns 8 output rows x 16 columns = 10 input rows; ragged tiles clamp their loads and skip their stores."""
    g = torch.Generator().manual_seed(KT + W)
    Cin = 96
    x = torch.randn(Cin, T + KT - 1, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, KT, 3, 3, generator=g, dtype=torch.float64)
    ref = torch.nn.functional.conv3d(torch.nn.functional.pad(x[None], (1, 1, 1, 1, 0, 0)), w)[0]  # [Cout, T, H, W]
    xp = torch.zeros(T + KT - 1, H + 2, W + 2, Cin, dtype=torch.float64)  # bordered channels-last frames
    xp[:, 1:-1, 1:-1] = x.permute(1, 2, 3, 0)
    A = torch.zeros(KT, 3, 3, 16, 32, dtype=torch.float64)  # weight fragments [kt][kw][chunk][n = (kh, co)][k]
    for kh in range(3):
        for co in range(Cout):
            A[:, :, :, kh * 4 + co] = w[co].permute(1, 3, 2, 0)[:, :, kh].reshape(KT, 3, 3, 32)  # [kt][kw][Cin -> (chunk, k)] of row kh
    out = torch.zeros(Cout, T, H, W, dtype=torch.float64)
    for t in range(T):
        for h0 in range(0, H, 8):
            for w0 in range(0, W, 16):
                cols = torch.clamp(torch.arange(w0, w0 + 16), max=W - 1)
                P = torch.zeros(10, 16, 16, dtype=torch.float64)  # [rho][n][pixel]
                for kt in range(KT):
                    for rho in range(10):
                        row = min(h0 + rho, H + 1)
                        for kw in range(3):
                            for c in range(3):
                                B = xp[t + kt, row, cols + kw, 32 * c: 32 * c + 32]  # [pixel][k]
                                P[rho] += A[kt, kw, c] @ B.t()
                for j in range(8):
                    if h0 + j >= H:
                        continue
                    o = P[j, 0:4] + P[j + 1, 4:8] + P[j + 2, 8:12]  # lane groups 0 / 1 / 2 hold kernel rows 0 / 1 / 2
                    ok = torch.arange(w0, w0 + 16) < W
                    out[:, t, h0 + j, w0: w0 + int(ok.sum())] = o[:Cout, : int(ok.sum())]
    assert torch.allclose(out, ref, rtol=1e-10, atol=1e-10), float((out - ref).abs().max())


def test_bench_bare_gpus_n_relaunches_itself_under_torch_distributed_run(monkeypatch, capsys):
    """`python bench.py --gpus 8` without a launcher (VERDICT r4 #1a): the process is replaced by the driver's own launch form - same script,
    same arguments, 127.0.0.1 rendezvous on a free port, dmabuf IPC kept - instead of raising; a box with fewer GPUs gets one parsable
    error line and exit code 2; a second re-launch (the launcher did not set WORLD_SIZE) is refused."""
    import json
    import os
    import sys

    import bench
    seen = {}

    def fake_execve(exe, cmd, env):
        seen.update(exe=exe, cmd=cmd, env=env)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execve", fake_execve)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("CE_BENCH_SELF_LAUNCHED", raising=False)
    monkeypatch.delenv("CE_BENCH_TEST_BACKEND", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit):
        bench._self_launch(8)
    cmd = seen["cmd"]
    assert seen["exe"] == sys.executable and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["CE_BENCH_SELF_LAUNCHED"] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # fewer GPUs than ranks: one JSON error line, exit code 2, no launch
    seen.clear()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as ei:
        bench._self_launch(8)
    assert ei.value.code == 2 and not seen
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 8 and "error" in line
    # ... except under the one-GPU test backend, where all ranks share GPU 0
    monkeypatch.setenv("CE_BENCH_TEST_BACKEND", "gloo")
    with pytest.raises(SystemExit):
        bench._self_launch(8)
    assert seen["cmd"][:3] == [sys.executable, "-m", "torch.distributed.run"]
    # a re-launched process that still has no rank environment must not loop
    monkeypatch.setenv("CE_BENCH_SELF_LAUNCHED", "1")
    with pytest.raises(RuntimeError):
        bench._self_launch(8)


def test_bench_median_time_bounded_sample():
    """cpu_baseline's sampling rule: median of the timed runs after the warm-up; with a budget, a warm-up call that alone exceeds a third of it IS the sample."""
    import time as _t

    import bench
    calls = []
    dt, ts = bench._median_time(lambda: calls.append(1), runs=3, warm=1)
    assert len(calls) == 4 and len(ts) == 3
    calls.clear()
    dt, ts = bench._median_time(lambda: (calls.append(1), _t.sleep(0.05)), runs=3, warm=1, budget_s=0.1)
    assert len(calls) == 1 and len(ts) == 1 and dt >= 0.05


def test_scaling_model_prediction_block():
    """tools/scaling_model.predict (what bench.py puts into `rccl.model_prediction`): every split of the ranks priced eager and captured; a captured
    loop exposes the k|v exchange it hides when eager, so it never ranks above its eager twin; the guidance-pair split is never offered captured."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import scaling_model as m
    for W in (2, 4, 8):
        p = m.predict(W)
        sps = p["steps_per_s"]
        assert p["choice"] in sps and sps[p["choice"]] == max(sps.values())
        assert not any("cfg-parallel" in k and "hipGraph" in k for k in sps)
        for k, v in sps.items():
            if k.endswith(", hipGraph"):
                assert v <= sps[k.replace(", hipGraph", ", eager")], (W, k)
        assert max(sps.values()) > p["one_gpu_steps_per_s"]
    assert m.predict(2)["choice"].startswith("cfg-parallel")       # one xGMI link between two ranks: split the guidance pair instead
    assert m.predict(8)["choice"] == "ulysses 8, B=2, eager"


def test_scaling_model_states_the_expected_wall_time_of_the_default_multi_gpu_line():
    """`rccl.model_prediction` (bench.py --gpus N at full size) carries the expected end-to-end wall time of the driver's
    default command, leg by leg, so that a first multi-GPU run that takes much longer reads as a hang and not as a slow run."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import scaling_model
    for W in (2, 4, 8):
        p = scaling_model.predict(W)
        w = p["expected_driver_wall_s"]
        assert 100 < w["total"] < 600 and abs(sum(w["legs"].values()) - w["total"]) < 1.0, w
        assert p["choice"] in p["steps_per_s"]
    assert scaling_model.predict(8)["expected_driver_wall_s"]["total"] < scaling_model.predict(2)["expected_driver_wall_s"]["total"]

