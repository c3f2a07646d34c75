"""Launch POLICY of the operators -- the decisions that are not arithmetic: how many sequence segments the chunk kernels
run concurrently, which backward implementation serves bf16, when a weight gradient is posed as a token-split batched GEMM,
how the channel mixer is padded.  One object (``POLICY``) holds the switches a test or a tool may flip; nothing here reads the
environment."""
from __future__ import annotations

import torch


class Policy:
    """``k2b_path``: "full" = K2b as three sweeps of the full-head kernel (bf16, Dk = Dv in {64,128,256}, falls back when the
    layout is not eligible), "sweeps" = always the generic kernel (lina_gla_chunk_bwd)."""
    k2b_path: str = "full"
    # 256 x 512 heads (expand_v = 2), chunk forward: True = both value column blocks in ONE launch, the two workgroups of a
    # head paired on one XCD (gla_chunk_full.hip, NCB = 2: q, k, g reach HBM once); False = one launch per column block
    # (rounds 2-5; still what the segment-parallel form and the backward do)
    dv512_one_launch: bool = True
    # train path: the stacked projection weight (K16) and the channel mixer's padded operands (K15) in ONE pass each over the
    # fp32 master weights; False = torch.cat / cast / fills / strided copies (rounds 3-6, ~190 more launches per step)
    one_pass_operands: bool = True
    # train path, the stacked projection (4160 output columns at L169: q | k | v | g | low-rank 16 | pad 48): the GEMM library runs
    # N = 4096 at 1.3 PFLOP/s and N = 4160 at 0.95 (and its weight gradient token-split at 259 us instead of 337 as one GEMM:
    # profiles/r06_inproj_gemm_split.txt) -- True = the forward and the weight gradient as a 256-aligned main product plus a
    # narrow tail product into / out of the same buffers; dX stays one GEMM
    split_stacked_gemm: bool = True
    # train path, channel mixer: the down-projection's operand in rows of 1536 (not 1408) elements so that its dX GEMM
    # ([M, 1024] x [1024, 1536]: 90-100 us; x [1024, 1408]: 112-119) runs on the width the GEMM library prefers
    wide_down_dx: bool = True


POLICY = Policy()


def _value_blocks(q, v, gk) -> int:
    """Dv = m * Dk with the L169 key width (``expand_v = 2``: 256 x 512 heads): the recurrence is independent per value
    column, so the call runs as m calls of the full-head 256 x 256 kernel on column blocks of v / o / the states (same q, k,
    g); the backward adds the blocks' dq, dk, dg.  Returns m (1 = no split)."""
    Dk, Dv = q.shape[-1], v.shape[-1]
    if q.dtype == torch.bfloat16 and gk.dtype == torch.bfloat16 and Dk == 256 and Dv > Dk and Dv % Dk == 0:
        return Dv // Dk
    return 1


def chunk_segments(n_heads_total: int, T: int) -> int:
    """Segments for the segment-parallel K2 (lina_gla_chunk_fwd_seg): enough to put ~256 workgroups on the chip
    when B*H is small, at least 256 tokens per segment; 1 = the plain kernel."""
    if n_heads_total >= 128 or T < 1024:
        return 1
    return max(1, min(256 // n_heads_total, T // 256, 16))


# --------------------------------------------------------------------------- projections of the train path
# The GEMMs stay on the vendor library (hipBLASLt through torch); what is ours is how the WEIGHT GRADIENT is posed to it.
# dW = dY^T X reduces over all B T tokens (32768 on config 5) into a small [out, in] tile grid: posed as one GEMM the
# library runs it at 300-580 TFLOP/s (1024x1024 / 1024x1365 / 2730x1024 outputs: 16-44 tiles for 256 CUs, profiles/
# r03_dw_gemm.txt); split over the token axis into a batched GEMM with fp32 partial products + one small sum it runs at
# 830-940 TFLOP/s, and dW comes out in fp32 (the master-weight dtype: no bf16 round trip, no cast kernel).
_LINEAR_SPLIT_MAX_OUT = 3 * 1024 * 1024        # [out, in] up to this many elements: split (above: enough tiles already)
_LINEAR_SPLIT_MIN_ROWS = 2048                  # tokens per split slice, at least


def _linear_split(rows, n_out, n_in):
    if n_out * n_in > _LINEAR_SPLIT_MAX_OUT:
        return 1
    for s in (8, 4, 2):
        if rows % s == 0 and rows // s >= _LINEAR_SPLIT_MIN_ROWS:
            return s
    return 1


_MLP_PAD = 128          # the hidden dimension of the channel mixer is padded to a multiple of this (+ the bias column)
