#!/usr/bin/env python3
"""Static check of the uniform-schedule kernels (csrc/ecgpu_ct.h) on their gfx950 ISA: no conditional branch and no memory
address may depend on the contents of a scalar or point record.

    python tools/ct_isa_check.py [--curve P256Params ...] [--kernels k_var_base_ct,k_fixed_base_ct,k_proj_sum_level] [--keep] [--self-test]
(k_proj_sum_level: the tree of complete additions ecgpu_lincomb_ct runs over the products of k_var_base_ct)

How: the translation unit is compiled to assembly (hipcc -S --offload-device-only; no GPU needed) and every selected kernel
goes through a forward taint analysis over its control-flow graph (register-precise, iterated to a fixed point):

  * sources   every vector memory load result (global_load / flat_load: scalar records, point records, identity flags, the
              per-lane table in HBM scratch — all of it is treated as secret), LDS reads once a tainted value has been written
              to LDS, scratch reloads of a slot that holds a tainted value (slots are tracked by their constant offset);
              s_load results are tainted only if their base address is (kernel arguments and .rodata constants are not)
  * flow      a destination is tainted when any source operand is, including vcc / scc / SGPR masks consumed by
              v_cndmask, v_addc, s_cselect ...; v_readlane / v_writelane spills are tracked per (register, lane)
  * sinks     (1) the condition of a conditional branch (scc, vcc, exec) or the target of s_setpc,
              (2) any write to exec from a tainted value (divergent control flow),
              (3) the address operands (vaddr / saddr / sbase) of every load, store, atomic and LDS access.
A violation is printed with its line; the exit status is the number of violations.  --self-test runs the same analysis on
the variable-time kernels (k_var_base / k_fixed_base), which MUST be reported: a checker that cannot see a digit-dependent
table gather would prove nothing.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "elliptic-curves_amd", "csrc")

REG = re.compile(r"^(v|s|a)(\d+)$|^(v|s|a)\[(\d+):(\d+)\]$")


def regs_of(tok):
    """register names an operand token stands for (empty for literals, modifiers, `off`)"""
    tok = tok.strip()
    if tok.startswith("-") or tok.startswith("|"):
        tok = tok.strip("-|")
    for pre in ("neg(", "abs(", "sext("):
        if tok.startswith(pre):
            tok = tok[len(pre):].rstrip(")")
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return ["vcc"]
    if tok in ("exec", "exec_lo", "exec_hi"):
        return ["exec"]
    if tok in ("m0", "scc"):
        return [tok]
    m = REG.match(tok)
    if not m:
        return []
    if m.group(1):
        return [m.group(1) + m.group(2)]
    return [m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]


class Ins:
    __slots__ = ("line", "text", "op", "ops", "offset", "long_target")

    def __init__(self, line, text):
        self.line, self.text = line, text
        body = text.split(";")[0].strip()
        parts = body.split(None, 1)
        self.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        toks = [t.strip() for t in rest.split(",")] if rest else []
        self.offset = 0
        ops = []
        for t in toks:
            sub = t.split()
            if not sub:
                continue
            ops.append(sub[0])
            for extra in sub[1:] + [sub[0]]:
                if extra.startswith("offset:"):
                    try:
                        self.offset = int(extra[7:], 0)
                    except ValueError:
                        pass
        self.ops = ops
        m = re.search(r"\((\.LBB\w+)-\.Lpost_getpc\d+\)&", body)
        self.long_target = m.group(1) if m else None


def parse_kernels(path):
    """name -> list of (label or None, Ins)"""
    kernels, cur, name = {}, None, None
    for ln, raw in enumerate(open(path), 1):
        line = raw.rstrip("\n")
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is None:
            continue
        s = line.strip()
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur.append((m.group(1), None))
            continue
        if s.startswith("."):
            if s.startswith(".Lfunc_end"):
                cur = None
            continue
        if re.match(r"^[a-z]", s):
            cur.append((None, Ins(ln, s)))
    return kernels


CARRY2 = ("v_add_co_", "v_addc_co_", "v_sub_co_", "v_subb_co_", "v_subrev_co_", "v_subbrev_co_", "v_mad_u64_u32", "v_mad_i64_i32",
          "v_div_scale")
NO_SCC = ("s_mov_", "s_movk_", "s_load_", "s_getpc_", "s_waitcnt", "s_nop", "s_cmov", "s_setpc", "s_swappc", "s_sleep",
          "s_barrier", "s_sendmsg", "s_branch", "s_cbranch", "s_endpgm", "s_bitset", "s_sext", "s_mul_i32", "s_mul_hi",
          "s_setreg", "s_getreg", "s_memtime", "s_memrealtime", "s_dcache", "s_icache")


class Taint:
    def __init__(self, body, name, verbose=False):
        self.name, self.verbose = name, verbose
        # basic blocks: a label starts one, a branch ends one
        self.blocks, self.label_to_block = [], {}
        cur, pending = [], []
        for label, ins in body:
            if label is not None:
                if cur:
                    self.blocks.append(cur)
                    cur = []
                pending.append(label)
                continue
            if pending and not cur:
                for l in pending:
                    self.label_to_block[l] = len(self.blocks)
                pending = []
            cur.append(ins)
            if ins.op.startswith("s_cbranch") or ins.op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                self.blocks.append(cur)
                cur = []
        if cur:
            self.blocks.append(cur)
        self.violations = []

    def succ(self, bi):
        blk = self.blocks[bi]
        if not blk:
            return [bi + 1] if bi + 1 < len(self.blocks) else []
        last = blk[-1]
        out = []
        if last.op == "s_endpgm":
            return out
        if last.op == "s_setpc_b64":          # a relaxed long branch: s_getpc / s_add_u32 (.LBBx - .Lpost_getpc) / s_addc / s_setpc
            for prev in reversed(blk[:-1]):
                if prev.long_target:
                    t = self.label_to_block.get(prev.long_target)
                    return [t] if t is not None else []
            return out
        if last.op == "s_branch":
            t = self.label_to_block.get(last.ops[0])
            return [t] if t is not None else []
        if last.op.startswith("s_cbranch"):
            t = self.label_to_block.get(last.ops[0])
            if t is not None:
                out.append(t)
        if bi + 1 < len(self.blocks):
            out.append(bi + 1)
        return out

    def step(self, ins, st, report):
        """transfer function; st: set of tainted names (mutated)"""
        op, ops = ins.op, ins.ops

        def t(names):
            return any(n in st for n in names)

        def setd(names, val):
            for n in names:
                if val:
                    st.add(n)
                else:
                    st.discard(n)

        def sink(kind, names):
            if report and t(names):
                self.violations.append((ins.line, kind, ins.text.split(";")[0].strip()))

        R = [regs_of(o) for o in ops]
        if op in ("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep") or op.startswith("s_dcache") or op.startswith("s_icache"):
            return
        if op == "s_branch":
            return
        if op.startswith("s_cbranch_scc"):
            sink("branch on tainted scc", ["scc"]); return
        if op.startswith("s_cbranch_vcc"):
            sink("branch on tainted vcc", ["vcc"]); return
        if op.startswith("s_cbranch_exec"):
            sink("branch on tainted exec", ["exec"]); return
        if op.startswith("s_setpc") or op.startswith("s_swappc"):
            sink("indirect jump through tainted register", sum(R, []))
            if report and not any(p.long_target for p in self.cur_block):
                self.violations.append((ins.line, "indirect jump that is not a relaxed long branch", ins.text.split(";")[0].strip()))
            return
        # ---- memory ----
        if op.startswith("global_load") or op.startswith("flat_load"):
            sink("load address from tainted register", sum(R[1:], []))
            setd(R[0], True); return
        if op.startswith("global_store") or op.startswith("flat_store"):
            sink("store address from tainted register", R[0] + (R[2] if len(R) > 2 else [])); return
        if op.startswith("global_atomic") or op.startswith("flat_atomic"):
            if len(R) >= 4 or (len(ops) >= 4):      # returning form: vdst, vaddr, vdata, saddr
                sink("atomic address from tainted register", R[1] + R[3] if len(R) > 3 else R[1])
                setd(R[0], True)
            else:
                sink("atomic address from tainted register", R[0] + (R[2] if len(R) > 2 else []))
            return
        if op.startswith("scratch_store"):
            sink("scratch address from tainted register", R[0] + (R[2] if len(R) > 2 else []))
            width = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4, "byte": 1, "short": 1}[op.split("_")[-1]]
            dyn = bool(R[0]) or (len(R) > 2 and bool(R[2]))
            for k in range(width):
                key = "scratch:*" if dyn else "scratch:%d" % (ins.offset + 4 * k)
                val = t(R[1][k:k + 1]) if k < len(R[1]) else t(R[1])
                if dyn:
                    if val:
                        st.add(key)
                else:
                    setd([key], val)
            return
        if op.startswith("scratch_load"):
            sink("scratch address from tainted register", R[1] + (R[2] if len(R) > 2 else []))
            dyn = bool(R[1]) or (len(R) > 2 and bool(R[2]))
            for k, d in enumerate(R[0]):
                if dyn:
                    val = any(x.startswith("scratch:") for x in st)
                else:
                    val = ("scratch:%d" % (ins.offset + 4 * k)) in st or "scratch:*" in st
                setd([d], val)
            return
        if op.startswith("ds_write") or op.startswith("ds_store"):
            sink("LDS address from tainted register", R[0])
            if t(sum(R[1:], [])):
                st.add("lds")
            return
        if op.startswith("ds_read") or op.startswith("ds_load"):
            sink("LDS address from tainted register", R[1])
            setd(R[0], "lds" in st); return
        if op.startswith("ds_"):                           # permutes, swizzles: data movement between lanes
            setd(R[0], t(sum(R[1:], []))); return
        if op.startswith("buffer_"):
            sink("buffer access (unexpected in these kernels)", ["exec"] if False else sum(R, []))
            if "load" in op:
                setd(R[0], True)
            return
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            sink("scalar load address from tainted register", sum(R[1:], []))
            setd(R[0], t(R[1])); return
        # ---- lane <-> scalar moves ----
        if op == "v_readlane_b32":
            lane = ops[2]
            key = "%s.%s" % (R[1][0], lane) if not regs_of(lane) else None
            if key is None:
                sink("v_readlane lane index from register", regs_of(lane))
                setd(R[0], t(R[1]) or any(x.startswith(R[1][0] + ".") for x in st))
            else:
                setd(R[0], key in st or R[1][0] in st)
            return
        if op == "v_writelane_b32":
            lane = ops[2]
            if regs_of(lane):
                if t(R[1]):
                    st.add(R[0][0])
            else:
                setd(["%s.%s" % (R[0][0], lane)], t(R[1]))
            return
        if op == "v_readfirstlane_b32":
            setd(R[0], t(R[1])); return
        # ---- compares ----
        if op.startswith("v_cmpx"):
            srcs = sum(R, [])
            if report and t(srcs):
                self.violations.append((ins.line, "exec written from tainted compare", ins.text.split(";")[0].strip()))
            return
        if op.startswith("v_cmp"):
            setd(R[0], t(sum(R[1:], []))); return
        if op.startswith("s_cmp") or op.startswith("s_bitcmp"):
            setd(["scc"], t(sum(R, []))); return
        # ---- 32 x 32 + 64 multiply-add: the low word of the result does not see the high word of the addend (the compiler
        # uses it as a 32-bit multiply-add with whatever the odd register of the pair holds) ----
        if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")) and len(R) >= 5 and len(R[0]) == 2:
            lo_src = R[2] + R[3] + R[4][:1]
            lo, hi = t(lo_src), t(R[2] + R[3] + R[4])
            setd(R[0][:1], lo); setd(R[0][1:], hi); setd(R[1], hi)
            return
        # ---- generic ALU ----
        ndst = 2 if op.startswith(CARRY2) else 1
        if op.startswith("s_and_saveexec") or op.startswith("s_or_saveexec") or op.startswith("s_andn2_saveexec"):
            src = R[1] + ["exec"]
            val = t(src)
            if report and t(R[1]):
                self.violations.append((ins.line, "exec written from tainted mask", ins.text.split(";")[0].strip()))
            setd(R[0], val); setd(["scc"], val); return
        dsts = sum(R[:ndst], [])
        srcs = sum(R[ndst:], [])
        if op.startswith("s_addc") or op.startswith("s_subb") or op.startswith("s_cselect"):
            srcs = srcs + ["scc"]
        if op.startswith("v_writelane"):
            srcs = srcs + dsts
        val = t(srcs)
        if "exec" in dsts and report and val:
            self.violations.append((ins.line, "exec written from tainted value", ins.text.split(";")[0].strip()))
        if op.startswith(("v_mov_b32_dpp", "v_mov_b32_sdwa")):
            pass
        # partial writes (SDWA / op_sel on 16-bit halves) keep the old contents: union with the destination
        if "_sdwa" in op or "dst_sel" in ins.text or "op_sel" in ins.text:
            val = val or t(dsts)
        setd(dsts, val)
        if op.startswith("s_") and not op.startswith(NO_SCC):
            setd(["scc"], val)

    def run(self):
        n = len(self.blocks)
        instate = [None] * n
        instate[0] = frozenset()
        work = [0]
        while work:
            bi = work.pop()
            st = set(instate[bi])
            for ins in self.blocks[bi]:
                self.step(ins, st, False)
            out = frozenset(st)
            for s in self.succ(bi):
                if s is None or s >= n:
                    continue
                new = out if instate[s] is None else instate[s] | out
                if new != instate[s]:
                    instate[s] = new
                    work.append(s)
        for bi in range(n):
            if instate[bi] is None:
                continue
            st = set(instate[bi])
            self.cur_block = self.blocks[bi]
            for ins in self.blocks[bi]:
                self.step(ins, st, True)
        return self.violations


def compile_asm(tu, curve, out):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DECGPU_CURVE=" + curve, "-S", "--offload-device-only",
           os.path.join(CSRC, tu), "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def summarize(body):
    c = {}
    for _, ins in body:
        if ins is not None and (ins.op.startswith("s_cbranch") or ins.op in ("s_branch", "s_setpc_b64")):
            c[ins.op] = c.get(ins.op, 0) + 1
    return ", ".join("%s x%d" % kv for kv in sorted(c.items()))


def check(asm, wanted, verbose=False):
    total = 0
    kernels = parse_kernels(asm)
    for name, body in kernels.items():
        if not any(w in name for w in wanted):
            continue
        n_ins = sum(1 for _, i in body if i is not None)
        if n_ins == 0:
            continue
        v = Taint(body, name, verbose).run()
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0] or name
        print("%-52s %6d instructions; branches: %s -> %s" % (short, n_ins, summarize(body) or "none",
                                                               "OK" if not v else "%d VIOLATIONS" % len(v)))
        seen = set()
        for line, kind, text in v:
            if (line, kind) in seen:
                continue
            seen.add((line, kind))
            if len(seen) <= 12:
                print("    line %d: %s: %s" % (line, kind, text))
        total += len(set((l, k) for l, k, _ in v))
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", action="append")
    ap.add_argument("--kernels", default="k_var_base_ct,k_fixed_base_ct,k_proj_sum_level")
    ap.add_argument("--asm", help="check an existing .s file instead of compiling")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--self-test", action="store_true", help="the variable-time kernels must be flagged")
    a = ap.parse_args()
    curves = a.curve or ["K256Params", "P256Params", "P384Params"]
    wanted = a.kernels.split(",")
    if a.asm:
        return check(a.asm, wanted)
    bad = 0
    tmp = tempfile.mkdtemp(prefix="ct_isa_")
    for c in curves:
        if a.self_test:
            flagged = 0
            for tu, ks in (("ecgpu_inst_var.hip", ["k_var_base"]), ("ecgpu_inst_base.hip", ["k_fixed_base"])):
                asm = os.path.join(tmp, "%s_%s.s" % (tu, c))
                compile_asm(tu, c, asm)
                print("== self-test, %s, %s (violations expected)" % (c, tu))
                flagged += 1 if check(asm, ks) > 0 else 0
            if flagged != 2:
                print("SELF-TEST FAILED: a variable-time kernel went unreported")
                bad += 1
        else:
            asm = os.path.join(tmp, "ct_%s.s" % c)
            compile_asm("ecgpu_inst_ct.hip", c, asm)
            print("== %s" % c)
            bad += check(asm, wanted)
    if not a.keep:
        for f in os.listdir(tmp):
            os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)
    else:
        print("assembly kept in", tmp)
    return bad


if __name__ == "__main__":
    sys.exit(min(main(), 255))
