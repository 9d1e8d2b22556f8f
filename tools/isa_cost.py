#!/usr/bin/env python
"""Weighted VALU issue cost of a stretch of gfx950 ISA (hipcc -S output), using the per-SIMD issue costs measured
by tools/ubench/valu_rate{2,3}.hip on MI355X (clocks per wave64 instruction, 4 waves/SIMD):
   full rate 2.4: v_fma/fmac/mul/add/sub_f32, v_mov_b32, v_add/sub_u32, v_and/or/xor_b32, v_lshrrev_b32
   half rate 4.3: v_cmp*, v_cndmask, v_min/max, v_lshlrev, v_bfe, v_cvt, v_readlane, v_mbcnt, v_add_co, v_mul_lo,
                  v_add3, v_lshl_add, v_mad_*, v_ffb*, v_bcnt, DPP, v_pk_*, v_med3, 64-bit ops
   quarter  8.3: v_exp, v_rcp, v_sqrt, v_rsq, v_log
usage: isa_cost.py file.s first_line last_line   (1-based, inclusive)"""
import re, sys, collections
FULL = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mac_f32", "v_madak_f32", "v_madmk_f32",
        "v_fmaak_f32", "v_fmamk_f32", "v_not_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32")
QUART = ("v_exp_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32", "v_rcp_iflag_f32", "v_sin_f32", "v_cos_f32")
def cost(op, line):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if "quad_perm" in line or "row_" in line or base.endswith("_dpp"): return 4.3
    if base in QUART: return 8.3
    if base in FULL: return 2.4
    return 4.3
def main():
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lines = open(f).read().split("\n")[a - 1:b]
    cnt = collections.Counter(); clk = collections.Counter()
    for l in lines:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        op = t.split()[0]
        if op.startswith("v_"):
            c = cost(op, t); cnt[op] += 1; clk[op] += c
        else:
            cnt[op.split("_")[0] + "_*"] += 1
    tot = sum(clk.values()); nv = sum(v for k, v in cnt.items() if k.startswith("v_"))
    print(f"VALU instructions {nv}, weighted issue clocks {tot:.0f} (= {tot/2.4:.0f} fma-equivalents)")
    for k, v in sorted(clk.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {k:28s} x{cnt[k]:4d}  {v:7.1f} clk")
    print("  other:", {k: v for k, v in cnt.items() if not k.startswith("v_")})
main()
