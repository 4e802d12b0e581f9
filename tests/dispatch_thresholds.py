"""The shape thresholds the C ABI's dispatch branches on -- ONE table, mirrored from the sources (round-5 verdict, item 3: two
silently wrong answers survived 2 - 3 rounds because the tested shapes were chosen by hand; both sat right behind a threshold).

Each entry: (name, axis, value, kinds, source file, snippet).  ``axis`` is what the branch compares -- "B" (batch size), "n2"
(= 2B, the rows of the factor forms' small chain), "n" (= B, BaM's small matrix), "D"; ``kinds`` are the update kinds whose
dispatch contains the branch; ``snippet`` is a literal piece of the source line that MUST be present in the file:
tests/test_thresholds_mirror.py (CPU) fails when a threshold moves in the source without this table following, and
tests/test_gpu_thresholds.py runs every kind at value - 2 .. value + 2 with a fresh, exactly sized context per case.
Kinds: "gsm" (dense GSM update), "gsmf" (factor-form GSM update), "bam" (dense BaM update), "bamf" (factor-form BaM update),
"potrf"."""

CSRC = "gsm-vi_amd/csrc/"

THRESHOLDS = [
    # ---- batch size of the dense GSM update ------------------------------------------------------------------------------------
    ("cov_sym staged rows 16", "B", 16, ("gsm",), CSRC + "gsmvi_fast.hip", "const int SB = B <= 16 ? 16 : (B <= 32 ? 32 : (B <= 64 ? 64 : 128));"),
    ("cov_sym staged rows 32", "B", 32, ("gsm",), CSRC + "gsmvi_fast.hip", "const int SB = B <= 16 ? 16 : (B <= 32 ? 32 : (B <= 64 ? 64 : 128));"),
    ("cov_sym staged rows 64", "B", 64, ("gsm",), CSRC + "gsmvi_fast.hip", "const int SB = B <= 16 ? 16 : (B <= 32 ? 32 : (B <= 64 ? 64 : 128));"),
    ("cov_sym run-time pass loop", "B", 128, ("gsm",), CSRC + "gsmvi_fast.hip", "if (B > 128) {          // round 6: any batch size"),
    ("panel rows per block 16 / 32", "B", 16, ("gsm", "bam"), CSRC + "gsmvi_abi.hip", "const int MT = nrows <= 16 ? 1 : (nrows <= 32 ? 2 : 4);"),
    ("panel row blocks of 64", "B", 64, ("gsm", "bam"), CSRC + "gsmvi_abi.hip", "const int zblocks = (nrows + 16 * MT - 1) / (16 * MT);"),
    # ---- BaM's B x B chain -----------------------------------------------------------------------------------------------------
    ("bam stats column groups", "n", 32, ("bam", "bamf"), CSRC + "gsmvi_bam.hip", "if (B <= 32) { if (Xh) STATS(4, true, (D + 63) / 64);"),
    ("bam one-launch chain", "n", 48, ("bam", "bamf"), CSRC + "gsmvi_bam_small.hip", "#define BAMQ_SN 48"),
    ("bam one-workgroup Newton-Schulz", "n", 64, ("bam", "bamf"), CSRC + "gsmvi_bam_small.hip", "whole iteration in ONE workgroup for n <= 64"),
    ("bam one-workgroup Cholesky", "n", 128, ("bam", "bamf"), CSRC + "gsmvi_bam_small.hip", "#define BAMS_NMAX 128"),
    ("bam low-rank update passes 96", "n", 48, ("bam",), CSRC + "gsmvi_bam.hip", "if (n2 <= 96) LRU(3, 32, false);"),
    ("bam low-rank update: run-time pass loop above 288 rows", "n", 144, ("bam",), CSRC + "gsmvi_bam.hip", "D % 2 == 0 && n2 > 288) {"),
    ("bam substitution 64 columns", "n", 160, ("bam",), CSRC + "gsmvi_bam.hip", "if (n <= 160) BFW(64); else if (n <= 320) BFW(32); else if (n <= 640) BFW(16); else BFW(8);"),
    ("bam substitution 32 columns", "n", 320, ("bam",), CSRC + "gsmvi_bam.hip", "if (n <= 160) BFW(64); else if (n <= 320) BFW(32); else if (n <= 640) BFW(16); else BFW(8);"),
    ("bam substitution 16 columns", "n", 640, ("bam",), CSRC + "gsmvi_bam.hip", "if (n <= 160) BFW(64); else if (n <= 320) BFW(32); else if (n <= 640) BFW(16); else BFW(8);"),
    ("bam largest batch", "n", 1024, ("bam",), CSRC + "gsmvi_bam_small.hip", "#define BAMS_NBIG 1024"),
    # ---- the 2B x 2B chain of the factor forms ---------------------------------------------------------------------------------
    ("factor update one staging pass", "n2", 32, ("gsmf", "bamf"), CSRC + "gsmvi_factor.hip", "const int npsel = n <= 32 ? 1 : (n <= 64 ? 2 : (n <= 128 ? 4 : 8));"),
    ("factor chain one workgroup (rider)", "n2", 64, ("gsmf", "bamf"), CSRC + "gsmvi_factor.hip", "const bool rider = ctx->tune_rider && n <= 64;"),
    ("factor chain one-workgroup factorisations", "n2", 128, ("gsmf", "bamf"), CSRC + "gsmvi_factor.hip", "} else if (n > 128) {"),
    ("factor forms largest chain", "n2", 256, ("gsmf", "bamf"), CSRC + "gsmvi_ctx.h", "#define GSMVI_FACTOR_NMAX 256"),
    ("factor lean path needs B % 16 above 64 rows", "n2", 96, ("gsmf", "bamf"), CSRC + "gsmvi_factor.hip", "(n <= 64 || (n <= 128 && B % 16 == 0))"),
    ("bamf given first block", "n2", 128, ("bamf",), CSRC + "gsmvi_bam.hip", "if (n <= 64 && ctx->tune_bam_basis != 3) {"),
    # ---- D ---------------------------------------------------------------------------------------------------------------------
    ("even D: tuned kernels", "D", 64, ("gsm", "gsmf", "bam", "bamf"), CSRC + "gsmvi_abi.hip", "const bool fast = !ctx->tune_no_fast && D % 2 == 0 && a_vec_ok"),
    ("D % 64: ragged panel / tile instances", "D", 128, ("gsm", "gsmf", "bam", "bamf"), CSRC + "gsmvi_fast.hip", "const bool rag = D % 64 != 0 || ncols % 16 != 0;"),
    ("D % 32: ragged covariance tiles", "D", 96, ("gsm",), CSRC + "gsmvi_fast.hip", "const bool rag = D % 32 != 0 || B != SB;"),
    ("side job of the product beside a 128-row chain", "D", 256, ("gsmf", "bamf"), CSRC + "gsmvi_fast.hip", "const int per = (px.sj_len + nwg - 1) / nwg;"),
    ("scalars: one element per thread", "D", 512, ("gsm",), CSRC + "gsmvi_fast.hip", "if (nt != 256 && nt != 512 && nt != 1024) nt = 512;"),
    ("scalars: two elements per thread / wide panels / unsplit rider product", "D", 1024, ("gsm", "gsmf", "bamf"), CSRC + "gsmvi_abi.hip", "ncols % 64 == 0 && ncols >= 1024 && D >= 1024 && D % 64 == 0"),
    ("potrf split solve from 24 tile rows", "D", 1536, ("potrf",), CSRC + "gsmvi_potrf.hip", "#define POTRF_SPLIT_M 24"),
    ("panel product keeps the next chunk's loads in flight from D = 2048", "D", 2048, ("gsm",), CSRC + "gsmvi_ctx.h",
     "int tune_panel_w4_min_D = 2048;"),
    ("unsplit rider product up to 2048", "D", 2048, ("gsm", "gsmf"), CSRC + "gsmvi_ctx.h", "int tune_rider_direct_max_D = 2048;"),
    ("persistent covariance kernel from 2048 items (D = 2850: 90 tile rows)", "D", 2848, ("gsm",), CSRC + "gsmvi_fast.hip", "if (n_items >= 2048 && SB <= 32 && !stamps && dbg == 0) {"),
    ("second stream from D = 3072", "D", 3072, ("gsmf", "bamf"), CSRC + "gsmvi_ctx.h", "int tune_fork_min_D = 3072;"),
    ("scalars: eight elements per thread, then the guarded kernel", "D", 4096, ("gsm",), CSRC + "gsmvi_fast.hip", "if (KC > 8 || ept > 8) return false;"),
]

# the batch size the D-axis cases run with, per kind (small: the oracle side must stay cheap at D = 4096) -- except where the
# threshold only exists for a larger chain
D_AXIS_BATCH = {"gsm": 8, "gsmf": 8, "bam": 8, "bamf": 8}
D_AXIS_BATCH_OVERRIDE = {
    ("side job of the product beside a 128-row chain", "gsmf"): 64,
    ("side job of the product beside a 128-row chain", "bamf"): 64,
    ("scalars: two elements per thread / wide panels / unsplit rider product", "gsm"): 64,     # 64-row panels: the wide kernel
    ("scalars: two elements per thread / wide panels / unsplit rider product", "gsmf"): 32,    # n = 64: the rider's product
    ("scalars: two elements per thread / wide panels / unsplit rider product", "bamf"): 32,
    ("unsplit rider product up to 2048", "gsmf"): 32,
    ("second stream from D = 3072", "gsmf"): 40,                                              # 64 < 2B <= 128
    ("second stream from D = 3072", "bamf"): 40,
}
# the dimension the B-axis cases run at, per kind (the factor forms need 2B <= D)
B_AXIS_DIM = {"gsm": 96, "bam": 64, "gsmf": 264, "bamf": 264}


def cases():
    """[(kind, D, B, why)] -- every kind of every threshold at value - 2 .. value + 2 on its axis, deduplicated."""
    seen, out = set(), []
    for name, axis, value, kinds, _file, _snip in THRESHOLDS:
        for kind in kinds:
            for off in (-2, -1, 0, 1, 2):
                if axis == "D":
                    D = value + off
                    if D > 1100 and off in (-1, 1):
                        continue                                   # (large D: the even neighbours only, each costs an O(D^2 B) oracle)
                    B = D_AXIS_BATCH_OVERRIDE.get((name, kind), D_AXIS_BATCH.get(kind, 8))
                else:
                    v = value + off
                    if axis == "n2":
                        if v % 2:
                            continue
                        B = v // 2
                    else:
                        B = v
                    D = B_AXIS_DIM[kind]
                if B < 1 or D < 2:
                    continue
                if kind in ("gsmf", "bamf") and (2 * B > D):
                    continue
                key = (kind, D, B)
                if key in seen:
                    continue
                seen.add(key)
                out.append((kind, D, B, f"{name} ({axis} = {value}{off:+d})"))
    return out
