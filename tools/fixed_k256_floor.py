#!/usr/bin/env python3
"""Instruction-level floor of the BASELINE headline kernel, k_fixed_base<K256Params> (2^20 scalars, configs[1]) — the model the
round-3 review asked for if the step could not be brought to 0.61 ms.

What the kernel must do per scalar, by construction (ecgpu_fixedmul.h; W = 26 comb: 10 windows):
    1 table entry taken as it is, 1 affine + affine addition (4M + 2S), 8 mixed XYZZ additions (8M + 2S each, nine
    reductions: the last two products of Y3 share one), the conversion XYZZ -> (X : Y : Z) (3M), the scalar decode and
    recode, 10 gathers of 64 B.
What a field operation must cost on gfx950 in issue slots (1 slot = one VOP3 / 64-bit issue = 4 cycles per wave64; a 32-bit
VOP1 / VOP2 is half a slot: profiles/r01/isa_issue_rates.txt), 9 x 29-bit limbs, 64-bit column accumulators:
    product columns     81 multiply-adds (a square: 45 + 9 half-slot doublings of the operand)
    reduction           8 high columns x (hand the upper half up: 1 multiply-add by 8, fold the lower half: 2) + the second-stage
                        column (3 + 4) + the top column folded in full (4)              = 32 multiply-adds
                        carry pass over 9 limbs: 8 x (64-bit shift + 64-bit add + mask)  = 8 x 2.5 slots
                        the last limb's overflow folded back: 4 multiply-adds + 3 masks + 2 shifts + 1 add ~ 7.5 slots
    linear steps        add / sub / negate-and-add of 9 limbs: 9 half slots; a 32-bit carry pass (`norm`): 9 x 3 half slots
The floor below multiplies these out; the measured side comes from profiles/roofline_consts.json (SQ_INSTS_VALU of the
kernel under rocprofv3 and the slot weights of its gfx950 ISA) and from the driver-timed kernel_ms.

    python tools/fixed_k256_floor.py [kernel_ms]        (default: 0.512, the mean of profiles/r04/k256_fused_sub_ab.txt)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rc = json.load(open(os.path.join(ROOT, "profiles", "roofline_consts.json")))["k_fixed_base<K256Params>"]
kernel_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.512
n = rc["units_per_launch"]

MUL_COLS, SQR_COLS = 81.0, 45.0 + 9 * 0.5
REDUCE = 32.0 + 8 * 2.5 + 7.5
LIN, NORM = 9 * 0.5, 9 * 1.5
M, S = MUL_COLS + REDUCE, SQR_COLS + REDUCE

# one mixed XYZZ addition (madd-2008-s): U2, S2, PP, PPP, Q, R^2, ZZ3, ZZZ3 each reduced, Y3 = R (Q - X3) - Y1 PPP as two
# products under one reduction; the three differences that feed a product (P, R, X3) leave the reduction of the product before
# them (F::mul_sub / F::sqr_sub, round 4): 9 multiply-adds by one each instead of a limb-wise subtraction + a carry pass; linear
# steps left: the three (multiple of p) - c, PPP + 2 Q, Q - X3, -Y1, the sign select of y
madd = (8 * MUL_COLS + 2 * SQR_COLS) + 9 * REDUCE + 3 * 9 + 8 * LIN
first = 4 * M + 2 * S + 9 + 6 * LIN + 2 * NORM                 # affine + affine (X3 through F::sqr_sub)
to_proj = 3 * M
decode = 8 + 8 * 1.0 + 24 * 0.5 + 10 * 6                       # byte swap, fold k -> n - k (8 words), 10 signed windows (shift / mask / sign)
unpack = 10 * 2 * 9 * 1.5                                      # 10 gathered entries, 2 coordinates: 8 words -> 9 limbs
floor_slots = 8 * madd + first + to_proj + decode + unpack

meas_insts = rc["insts_valu"] * 64 / n
meas_slots = meas_insts * rc["slots_per_inst"]
SIMDS, CLK = 1024, 2.4e9
slot_s = 4 / CLK
t_floor_issue = floor_slots * n / 64 / SIMDS * slot_s * 1e3
t_meas_issue = meas_slots * n / 64 / SIMDS * slot_s * 1e3
top = rc["isa_top"]
print("k_fixed_base<K256Params>, %d scalars per launch, kernel %.3f ms" % (n, kernel_ms))
print()
print("by construction, issue slots per scalar:")
print("  field mul %.1f  (81 product multiply-adds + %.1f reduction)   field sqr %.1f" % (M, REDUCE, S))
print("  mixed XYZZ addition %.0f   x 8 = %.0f" % (madd, 8 * madd))
print("  affine + affine %.0f, XYZZ -> projective %.0f, scalar decode + window recode %.0f, entry unpacking %.0f" % (first, to_proj, decode, unpack))
print("  FLOOR                                   %8.0f slots per scalar" % floor_slots)
print("measured (rocprofv3 SQ_INSTS_VALU x ISA slot weights, profiles/roofline_consts.json):")
print("  %.0f VALU instructions x %.4f slots   = %8.0f slots per scalar   (%.1f %% above the floor)" % (
    meas_insts, rc["slots_per_inst"], meas_slots, 100 * (meas_slots / floor_slots - 1)))
print("  static ISA of the kernel: %d v_mad_u64_u32, %d v_mov_b32 (%.1f %% of the slots), %d 64-bit shifts + %d 64-bit adds" % (
    top["v_mad_u64_u32"], top["v_mov_b32_e32"], 100 * top["v_mov_b32_e32"] * 0.5 / (rc["static_valu_instructions"] * rc["slots_per_inst"]),
    top["v_lshrrev_b64"], top["v_lshl_add_u64"]))
print()
print("time at 100 %% issue (1024 SIMDs, one slot per 4 cycles, 2.4 GHz):  floor %.3f ms, executed code %.3f ms" % (t_floor_issue, t_meas_issue))
print("measured kernel %.3f ms  ->  issue utilisation %.2f at 2.4 GHz (frac), %.2f of the cycles the chip had (frac_cycles_pmc: the clock" % (
    kernel_ms, t_meas_issue / kernel_ms, rc["insts_valu"] * rc["slots_per_inst"] * 4 / (rc["gui_cycles"] / 8 * SIMDS)))
print("  settles at ~2.1 GHz under this load)")
util_best = 0.90          # what the gather-free ladders reach (k_var_base<P384Params>: frac_cycles_pmc 0.90)
clk_eff = t_meas_issue / kernel_ms / (rc["insts_valu"] * rc["slots_per_inst"] * 4 / (rc["gui_cycles"] / 8 * SIMDS))
print()
print("what is left, in the kernel's own terms:")
print("  (a) the executed code is %.1f %% above the by-construction floor: register moves (%d static v_mov_b32) and the operand" % (
    100 * (meas_slots / floor_slots - 1), top["v_mov_b32_e32"]))
print("      marshalling of 64-bit accumulator pairs outside the reductions (round 5 wrote the REDUCTION by hand, csrc/ecgpu_k256_reduce_asm.h:")
print("      round 4 stood at +10.5 % with 370 moves; what is left sits in the affine + affine addition, the conversions and the recoding)")
print("  (b) the kernel issues in %.0f %% of its cycles; the gather-free ladders reach 90 %%.  The difference is the 10 random 64-byte" % (
    100 * rc["insts_valu"] * rc["slots_per_inst"] * 4 / (rc["gui_cycles"] / 8 * SIMDS)))
print("      gathers per scalar from a 21.5 GB table (measured: 2^20 copies of ONE scalar run 12 % faster at every table size,")
print("      DESIGN.md section 8); narrower tables need more additions and lose more (W = 16: +45 %)")
best = kernel_ms * floor_slots / meas_slots
print("  kernel with (a) closed completely: %.3f ms; with (a) and the gathers free: %.3f ms" % (
    best, best * (rc["insts_valu"] * rc["slots_per_inst"] * 4 / (rc["gui_cycles"] / 8 * SIMDS)) / util_best))
print("  step = kernel + k_normalize (0.121 ms: one inversion per lane, latency-bound; the workgroup-shared inversion of round 4")
print("  measured 0.121-0.133, profiles/r04/normalize_wg_dead_end.txt): %.3f ms with (a) closed; measured step 0.608-0.626 ms" % (best + 0.121))
