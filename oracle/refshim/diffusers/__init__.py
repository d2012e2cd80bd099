"""Stand-in for the absent ``diffusers==0.35.2`` wheel (TEST INFRASTRUCTURE ONLY).

Purpose: let oracle/gen_golden.py import and execute the reference's *own*
/root/reference/chronoedit_diffusers/transformer_chronoedit.py in this container, so
its block structure, dtype islands, attention processor and temporal-skip RoPE are
what produce the golden vectors under tests/golden/.  Only the leaf modules the
reference imports from diffusers (transformer_chronoedit.py:23-32) are restated here,
from the published behaviour of diffusers 0.35.2; those leaves are therefore NOT
pinned by the real wheel ("parity unpinned" for them).
Never imported by the product package.
"""
__version__ = "0.35.2-shim"
