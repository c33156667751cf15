#!/usr/bin/env python3
"""Builds profiles/roofline_consts.json — the executed-work constants bench.py prices its kernels with.

    python tools/roofline_consts.py <dir> [<dir> ...]   (<dir>/pmc_<workload>.json as written by `tools/gpu_run.sh <tag>
                                                  pmc:<workload>`; a kernel's counters are taken from its OWN workload's file —
                                                  the same kernel also runs at other sizes in the setup of other workloads)

For every headline kernel:
    insts_valu        VALU wave-instructions per launch                 (rocprofv3 --pmc SQ_INSTS_VALU)
    fetch_bytes / write_bytes   HBM traffic per launch                  (FETCH_SIZE / WRITE_SIZE, KiB -> bytes; separate passes)
    gui_cycles        GRBM_GUI_ACTIVE per launch (all 8 XCDs summed)
    slots_per_inst    mean issue cost of the kernel's VALU instructions in units of one v_mad_u64_u32 slot (4 cycles):
                      static histogram of the gfx950 ISA, VOP1/VOP2 (_e32) encodings = 1/2 slot, everything else 1
                      (measured costs: profiles/r01/isa_issue_rates.txt)
    mad_share         share of the VALU instructions that are v_mad_u64_u32
The ISA comes from compiling the kernel's translation unit with -S here (no GPU needed)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elliptic-curves_amd", "csrc")

# bench kernel name -> (instantiation group, curve struct, mangled-name substring, units per launch, workload)
KERNELS = {
    # (<C, true> / <C, 0, true>: the quad-major hand-over between the two, the default since round 6; tools/pmc_summary.py drops the flag)
    "k_fixed_base<K256Params>": ("base", "K256Params", "k_fixed_baseINS_10K256ParamsELb1E", 1 << 20, "fixed_k256"),
    "k_normalize<K256Params, 0>": ("base", "K256Params", "k_normalizeINS_10K256ParamsELi0ELb1E", 1 << 20, "fixed_k256"),
    "k_var_base<P256Params>": ("var", "P256Params", "k_var_baseINS_10P256ParamsELb0E", 1 << 20, "var_p256"),     # <C, false>: the plain ladder
    "k_var_base<P384Params>": ("var", "P384Params", "k_var_baseINS_10P384ParamsELb0E", 1 << 20, "var_p384"),
    "k_var_base<K256Params>": ("var", "K256Params", "k_var_baseINS_10K256ParamsELb1E", 1 << 20, "recover_k256"),   # <C, true>: adds its product to a G;   # b R of a G + b R: 2^20 launches per call
    "k_msm_accumulate<K256Params>": ("msm", "K256Params", "k_msm_accumulate", 1 << 24, "msm_k256"),
    "k_var_base_ct<P256Params>": ("ct", "P256Params", "k_var_base_ct", 1 << 20, "var_p256_ct"),
    "k_fixed_base_ct<K256Params>": ("ct", "K256Params", "k_fixed_base_ct", 1 << 20, "fixed_k256_ct"),
    "k_var_base_ct<K256Params>": ("ct", "K256Params", "k_var_base_ct", 1 << 20, "lincomb_ct_k256"),
}
# per-group compile flags of elliptic-curves_amd/Makefile (FLAGS_<group>: none at present)
GROUP_FLAGS = {}
HALF_SLOT_EXCEPT = ("_co_",)          # carry-producing / -consuming VOP2 ops were measured at the full cost

def isa_histogram(group, curve, substr):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DECGPU_CURVE=" + curve, "-S",
                               "--cuda-device-only", "-o", out, os.path.join(CSRC, "ecgpu_inst_%s.hip" % group)] + GROUP_FLAGS.get(group, []),
                              stderr=subprocess.DEVNULL)
        txt = open(out).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        if substr not in m.group(1):
            continue
        ops = [l.split()[0] for l in m.group(2).splitlines() if re.match(r"^\s+v_", l)]
        c = collections.Counter(ops)
        total = sum(c.values())
        # (inline-asm instructions print without an encoding suffix: the v_cmp / v_cndmask of ecgpu_ctmul.h are 32-bit encodings)
        # (... and the masks / moves / 32-bit adds of the k256 reduction blocks, ecgpu_k256_reduce_asm.h)
        half = sum(v for k, v in c.items() if (k.endswith("_e32") or k in ("v_cndmask_b32", "v_cmp_ne_u32", "v_and_b32", "v_mov_b32", "v_add_u32"))
                   and not any(x in k for x in HALF_SLOT_EXCEPT))
        mad = sum(v for k, v in c.items() if k.startswith("v_mad_u64_u32"))
        return {"static_valu": total, "slots_per_inst": (total - half / 2) / total, "mad_share": mad / total,
                "top": dict(c.most_common(8))}
    raise SystemExit("kernel %s not found in the ISA of %s/%s" % (substr, group, curve))


def main():
    dirs = sys.argv[1:]          # several directories: a workload's counters come from the LAST one that holds its file
    out = {}
    for name, (group, curve, substr, units, workload) in KERNELS.items():
        found = [os.path.join(d, "pmc_%s.json" % workload) for d in dirs if os.path.exists(os.path.join(d, "pmc_%s.json" % workload))]
        if not found:
            continue
        path = found[-1]
        with open(path) as f:
            rec = json.load(f).get(name)
        if not rec or "SQ_INSTS_VALU" not in rec:
            continue
        rec["_source"] = os.path.relpath(os.path.abspath(path), ROOT)
        h = isa_histogram(group, curve, substr)
        out[name] = {
            "workload": workload, "units_per_launch": units, "insts_valu": rec["SQ_INSTS_VALU"],
            "slots_per_inst": round(h["slots_per_inst"], 4), "mad_share": round(h["mad_share"], 4),
            "static_valu_instructions": h["static_valu"], "isa_top": h["top"],
            # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB (guide: MI355X_MICROARCH.md, HBM section)
            "fetch_bytes": rec["FETCH_SIZE"] * 1024 if "FETCH_SIZE" in rec else None,
            "write_bytes": rec["WRITE_SIZE"] * 1024 if "WRITE_SIZE" in rec else None,
            "gui_cycles": rec.get("GRBM_GUI_ACTIVE"), "sq_busy_cycles": rec.get("SQ_BUSY_CYCLES"),
            "sq_wave_cycles": rec.get("SQ_WAVE_CYCLES"), "waves": rec.get("SQ_WAVES"),
            "source": rec["_source"] + " + ISA histogram (tools/roofline_consts.py)",
        }
        print("%-32s insts %.4g  slots/inst %.3f  mad %.3f  fetch %s write %s" % (
            name, out[name]["insts_valu"], h["slots_per_inst"], h["mad_share"], out[name]["fetch_bytes"], out[name]["write_bytes"]))
    dst = os.path.join(ROOT, "profiles", "roofline_consts.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
