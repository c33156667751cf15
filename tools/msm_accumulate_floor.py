#!/usr/bin/env python3
"""Instruction-level floor of k_msm_accumulate<K256Params>, the dominant kernel of BASELINE configs[3] (2^24-term k256 MSM) —
the companion of tools/fixed_k256_floor.py, same slot accounting (1 slot = one VOP3 / 64-bit issue = 4 cycles per wave64; a 32-bit
VOP1 / VOP2 is half a slot).

By construction the kernel performs one mixed XYZZ addition (8M + 2S, nine reductions, three differences folded into their
products' reductions) per (term, window) entry — 16 windows of 16 bits for a folded 255-bit scalar —, unpacks the gathered point
(2 coordinates: 8 words -> 9 limbs), reads a quarter of a 16-byte index load, and writes a partial sum per bucket boundary
(negligible at 2^24 terms: one per ~512 entries).  The measured side: SQ_INSTS_VALU of the kernel under rocprofv3 and the slot weights of
its gfx950 ISA (profiles/roofline_consts.json), the kernel time from the bench line.

    python tools/msm_accumulate_floor.py [kernel_ms]       (default 14.3: profiles/r04/bench_default_closing*.json)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rc = json.load(open(os.path.join(ROOT, "profiles", "roofline_consts.json")))["k_msm_accumulate<K256Params>"]
kernel_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 14.3
terms = rc["units_per_launch"]
adds = terms * 16                                   # (term, window) entries; a zero digit (2^-16 of them) adds nothing

MUL_COLS, SQR_COLS = 81.0, 45.0 + 9 * 0.5
REDUCE = 32.0 + 8 * 2.5 + 7.5
LIN = 9 * 0.5
madd = (8 * MUL_COLS + 2 * SQR_COLS) + 9 * REDUCE + 3 * 9 + 8 * LIN
unpack = 2 * 9 * 1.5
loop = 12.0                                         # index stream (a 16-byte load per four entries), address of the gather, bucket-boundary test
floor_slots = madd + unpack + loop

meas_insts = rc["insts_valu"] * 64 / adds
meas_slots = meas_insts * rc["slots_per_inst"]
SIMDS, CLK = 1024, 2.4e9
t_floor = floor_slots * adds / 64 / SIMDS * 4 / CLK * 1e3
t_meas = meas_slots * adds / 64 / SIMDS * 4 / CLK * 1e3
cyc = rc["insts_valu"] * rc["slots_per_inst"] * 4 / (rc["gui_cycles"] / 8 * SIMDS)
print("k_msm_accumulate<K256Params>, %d terms x 16 windows = %.3g mixed additions per launch, kernel %.2f ms" % (terms, adds, kernel_ms))
print()
print("by construction, issue slots per addition:  mixed XYZZ addition %.0f + point unpacking %.0f + loop %.0f = %.0f" % (madd, unpack, loop, floor_slots))
print("measured: %.0f VALU instructions x %.4f slots = %.0f slots per addition  (%.1f %% above the floor)" % (
    meas_insts, rc["slots_per_inst"], meas_slots, 100 * (meas_slots / floor_slots - 1)))
print("  static ISA of the mixed addition (tools/k256_madd_isa_diff.py, profiles/r05/k256_madd_isa_diff.txt): 1,504 instructions, 1,071 of them")
print("  v_mad_u64_u32 (8 x 81 + 2 x 45 products, 9 reductions of 34, 27 for the folded differences), 27 v_mov_b32 (round 4: 1,629 / 91): the")
print("  reduction is a hand-scheduled assembly block (csrc/ecgpu_k256_reduce_asm.h); a step's stores and loads are all issued at its head")
print()
print("time at 100 %% issue (1024 SIMDs, one slot per 4 cycles, 2.4 GHz):  floor %.2f ms, executed code %.2f ms" % (t_floor, t_meas))
print("measured kernel %.2f ms  ->  issue utilisation %.2f at 2.4 GHz (frac), %.2f of the cycles the chip had (frac_cycles_pmc)" % (kernel_ms, t_meas / kernel_ms, cyc))
print()
print("round 4 executed 1,468 slots per addition (+6.1 % over this floor) in 14.3 ms; what closed the gap in round 5: the reduction in assembly")
print("(-5.4 % static slots of the addition) and the loop's bookkeeping (no register copies at the back edge, index quads and bucket ends")
print("fetched ahead).  What is left is the clock (~2.0 GHz under this load against the 2.4 the roof is drawn at) and 15 % of the cycles in")
print("which no wave of a SIMD can issue.  HBM traffic: %.1f GB fetched per launch against %.1f GB algorithmic (64 B point + 4 B index per" % (
    rc["fetch_bytes"] / 1e9, adds * 68 / 1e9))
print("addition) = %.2f TB/s of 8: the kernel is bound by its vector instructions, not by its gathers" % (rc["fetch_bytes"] / 1e9 / kernel_ms))
