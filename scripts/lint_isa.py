#!/usr/bin/env python3
"""ISA lint of libvscmi.so's gfx950 code objects: the hazards the compiler cannot see inside inline assembly.

Round 4 found one the hard way (vsc2022_amd/csrc/sim_i8p.hip, `bload_asm`): an inline-asm `buffer_load` whose
`soffset` SGPR had just been reloaded from a VGPR lane with `v_readlane` -- a VALU-written SGPR needs 5 wait states
before a VMEM instruction reads it, the hazard recognizer does not look inside the assembly, and one hit in 10^7 went
missing in 16 % of the runs.  The stream loads of the pre-filter kernels are also ASYNCHRONOUS behind the compiler's
back: their destination registers must not be touched (used, copied, spilled) before the hand-placed `s_waitcnt
vmcnt(N)` that covers them -- register allocation decides whether that holds.  This script checks both on
the machine code that actually ships, for every kernel of every code object:

  A  VALU writes SGPR / VCC (v_readlane, v_readfirstlane, v_cmp, carry-outs)  ->  VMEM reads it: 5 wait states
  B  VALU writes SGPR / VCC  ->  v_readlane / v_writelane uses it as the lane select: 4 wait states
  C  SALU writes M0  ->  LDS-DMA (`buffer_load ... lds`), s_sendmsg: 1 wait state
  D  a VGPR / AGPR that an outstanding VMEM load will write is read or written before an `s_waitcnt vmcnt(N)` has
     retired that load (loads and stores retire in issue order on gfx9's single vmcnt counter)

(wait states: every instruction issued in between counts 1, `s_nop N` counts N + 1; rules from the gfx90a / gfx940 ISA
guides' "manually inserted wait states" tables, the ones LLVM's GCNHazardRecognizer implements for compiler-generated
code).  Control flow is followed over basic blocks (the register ring of the stream crosses loop back edges).

    python scripts/lint_isa.py [path/to/libvscmi.so] [--verbose] [--spills]

Exit status 1 when a violation is found.  `tests/test_isa_lint.py` runs it on the shipped library (CPU suite) and on two
hand-assembled snippets that must fail / pass; the Makefile runs it after every link.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import deque

LLVM = os.environ.get("VSC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

_REG = re.compile(r"\b([vsa])(?:\[(\d+):(\d+)\]|(\d+)\b)")
_SPECIAL = re.compile(r"\b(vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|m0)\b")
_ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<([^>+]+)\+0x([0-9a-fA-F]+)>\s*$")
_FUNC = re.compile(r"^([0-9a-f]+) <([^>]+)>:")

VMEM_PREFIX = ("buffer_", "global_", "flat_", "scratch_", "tbuffer_")


def regs_of(text):
    """register names mentioned in an operand string: {'v12', 's3', 'a7', 'vcc', 'm0', ...}"""
    out = set()
    for m in _REG.finditer(text):
        kind = m.group(1)
        if m.group(4) is not None:
            out.add(f"{kind}{int(m.group(4))}")
        else:
            for k in range(int(m.group(2)), int(m.group(3)) + 1):
                out.add(f"{kind}{k}")
    for m in _SPECIAL.finditer(text):
        out.add(m.group(1).split("_")[0])
    return out


class Inst:
    __slots__ = ("addr", "mnem", "ops", "text", "target", "line", "facts")

    def __init__(self, addr, mnem, ops, text, target, line):
        self.addr, self.mnem, self.ops, self.text, self.target, self.line = addr, mnem, ops, text, target, line
        self.facts = None


def parse_disassembly(path):
    """{function name: [Inst]} from `llvm-objdump -d` output"""
    funcs, cur, base = {}, None, {}
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            m = _FUNC.match(raw)
            if m:
                cur = m.group(2)
                funcs[cur] = []
                base[cur] = int(m.group(1), 16)
                continue
            if cur is None or not raw.startswith("\t"):
                continue
            body, _, comment = raw.partition("//")
            am = _ADDR.search("//" + comment)
            if not am:
                continue
            body = body.strip()
            if not body:
                continue
            parts = body.split(None, 1)
            mnem = parts[0]
            ops = parts[1] if len(parts) > 1 else ""
            target = None
            tm = _TARGET.search(raw.strip())
            if tm and mnem.startswith(("s_cbranch", "s_branch")):
                target = (tm.group(1), int(tm.group(2), 16))
            funcs[cur].append(Inst(int(am.group(1), 16), mnem, ops, body, target, ln))
    return funcs, base


def split_operands(ops):
    out, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def is_vmem(i):
    return i.mnem.startswith(VMEM_PREFIX)


def is_valu(i):
    return i.mnem.startswith("v_")


def valu_scalar_writes(i):
    """SGPRs / vcc a VALU instruction writes"""
    ops = split_operands(i.ops)
    if not ops:
        return set()
    w = {r for r in regs_of(ops[0]) if r[0] == "s" or r == "vcc"}
    if len(ops) > 1 and re.search(r"_co_|v_mad_u64|v_mad_i64|v_div_scale|v_addc|v_subb", i.mnem):
        w |= {r for r in regs_of(ops[1]) if r[0] == "s" or r == "vcc"}
    if i.mnem.startswith("v_cmpx"):
        w.add("exec")
    return w


def vmem_dest(i):
    """VGPRs / AGPRs a VMEM instruction will write asynchronously (empty: store, LDS-DMA, atomic without return)"""
    if not is_vmem(i):
        return frozenset()
    toks = i.text.split()
    if " lds" in " " + i.ops + " " or toks[-1] == "lds":
        return frozenset()
    ops = split_operands(i.ops)
    if "_load" in i.mnem:
        return frozenset(r for r in regs_of(ops[0]) if r[0] in "va")
    if "atomic" in i.mnem and (" glc" in i.text or " sc0" in i.text):
        return frozenset(r for r in regs_of(ops[0]) if r[0] in "va")
    return frozenset()


def wait_states(i):
    if i.mnem == "s_nop":
        try:
            return int(i.ops.strip(), 0) + 1
        except ValueError:
            return 1
    return 1


def vmcnt_of(i):
    """N of `s_waitcnt vmcnt(N)`, None when the instruction does not wait on vmcnt"""
    if i.mnem != "s_waitcnt":
        return None
    m = re.search(r"vmcnt\((\d+)\)", i.ops)
    if m:
        return int(m.group(1))
    if re.fullmatch(r"\s*(0x[0-9a-fA-F]+|\d+)\s*", i.ops):  # raw immediate: vmcnt = bits [3:0] + [15:14] (gfx9)
        v = int(i.ops.strip(), 0)
        return (v & 0xF) | (((v >> 14) & 0x3) << 4)
    return None


class Facts:
    """everything the walk needs about one instruction, parsed once"""
    __slots__ = ("vregs", "sregs", "vmem", "lds_dma", "dest", "ws", "vmcnt", "valu_w", "m0_w", "lane_sel", "sendmsg")

    def __init__(self, ins):
        mentioned = regs_of(ins.ops)
        self.vmem = is_vmem(ins)
        self.dest = vmem_dest(ins)
        if self.vmem and self.dest:
            # a load may overwrite the destination of an older outstanding load (they retire in order): only its
            # address / data operands count as "touched"
            mentioned = regs_of(",".join(split_operands(ins.ops)[1:]))
        self.vregs = frozenset(r for r in mentioned if r[0] in "va")
        self.sregs = frozenset(r for r in mentioned if r[0] == "s" or r == "vcc")
        self.lds_dma = self.vmem and (ins.text.rstrip().endswith(" lds") or " lds " in ins.text + " ")
        self.ws = wait_states(ins)
        self.vmcnt = vmcnt_of(ins)
        self.valu_w = frozenset(valu_scalar_writes(ins)) if is_valu(ins) else frozenset()
        self.m0_w = False
        if ins.mnem.startswith("s_") and not ins.mnem.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch")):
            ops = split_operands(ins.ops)
            self.m0_w = bool(ops) and "m0" in regs_of(ops[0])
        self.lane_sel = frozenset()
        if ins.mnem.startswith(("v_readlane", "v_writelane")):
            ops = split_operands(ins.ops)
            self.lane_sel = frozenset(regs_of(ops[2])) if len(ops) > 2 else frozenset()
        self.sendmsg = ins.mnem.startswith("s_sendmsg")


def lint_function(name, insts, base_addr, verbose=False, max_visits=12):
    """-> list of violation strings"""
    if not insts:
        return []
    index_of = {ins.addr: k for k, ins in enumerate(insts)}
    # basic-block leaders
    leaders = {0}
    for k, ins in enumerate(insts):
        if ins.mnem.startswith(("s_cbranch", "s_branch")):
            if ins.target is not None:
                t = index_of.get(base_addr + ins.target[1])
                if t is not None:
                    leaders.add(t)
            if k + 1 < len(insts):
                leaders.add(k + 1)
        if ins.mnem in ("s_endpgm", "s_setpc_b64") and k + 1 < len(insts):
            leaders.add(k + 1)
    leaders = sorted(leaders)
    block_end = {b: (leaders[n + 1] if n + 1 < len(leaders) else len(insts)) for n, b in enumerate(leaders)}
    violations = {}

    def report(ins, rule, msg):
        violations.setdefault((ins.addr, rule), f"{name} @{ins.addr:#x} (disassembly line {ins.line}): [{rule}] {msg}: {ins.text}")

    # state: (recent, pending); recent = tuple of (reg, kind, age) of VALU scalar writes / SALU m0 writes younger than
    # 5 wait states; pending = tuple of frozenset (destination registers of outstanding VMEM operations, oldest first)
    start = ((), ())
    seen = {b: set() for b in leaders}
    visits = {b: 0 for b in leaders}
    work = deque([(0, start)])
    capped = False
    while work:
        b, (recent, pending) = work.popleft()
        key = (recent, pending)
        if key in seen[b]:
            continue
        if visits[b] >= max_visits:
            capped = True
            continue
        seen[b].add(key)
        visits[b] += 1
        recent = list(recent)
        pending = list(pending)
        k = b
        end = block_end[b]
        fall = True
        while k < end:
            ins = insts[k]
            f = ins.facts
            if f is None:
                f = ins.facts = Facts(ins)
            # ---- D: touching a register an outstanding load will still write
            if pending and f.vregs:
                for p in pending:
                    if p and (f.vregs & p):
                        report(ins, "D", f"{sorted(f.vregs & p)} is the destination of a VMEM load that no s_waitcnt has retired yet")
                        break
            # ---- A / B / C
            if recent:
                if f.vmem:
                    for reg, kind, age in recent:
                        if kind == "valu" and reg in f.sregs and age < 5:
                            report(ins, "A", f"{reg} was written by a VALU instruction {age} wait state(s) earlier (needs 5)")
                        if kind == "m0" and f.lds_dma and age < 1:
                            report(ins, "C", "m0 was written by the previous SALU instruction (needs 1 wait state)")
                elif f.lane_sel:
                    for reg, kind, age in recent:
                        if kind == "valu" and reg in f.lane_sel and age < 4:
                            report(ins, "B", f"lane select {reg} was written by a VALU instruction {age} wait state(s) earlier (needs 4)")
                elif f.sendmsg:
                    for reg, kind, age in recent:
                        if kind == "m0" and age < 1:
                            report(ins, "C", "m0 was written by the previous SALU instruction (needs 1 wait state)")
            # ---- advance the state past this instruction
            if recent:
                recent = [(r, kd, a + f.ws) for (r, kd, a) in recent if a + f.ws < 5]
            for r in f.valu_w:
                recent.append((r, "valu", 0))
            if f.m0_w:
                recent.append(("m0", "m0", 0))
            if f.vmcnt is not None:
                while len(pending) > f.vmcnt:
                    pending.pop(0)
            if f.vmem:
                pending.append(f.dest)
                if len(pending) > 63:
                    pending.pop(0)
            # operations older than the oldest outstanding LOAD protect nothing
            while pending and not pending[0]:
                pending.pop(0)
            # ---- control flow
            if ins.mnem in ("s_endpgm", "s_setpc_b64"):
                fall = False
                break
            if ins.mnem.startswith(("s_cbranch", "s_branch")):
                state = (tuple(sorted(recent)), tuple(pending))
                if ins.target is not None:
                    t = index_of.get(base_addr + ins.target[1])
                    if t is not None:
                        work.append((t, state))
                if ins.mnem.startswith("s_branch"):
                    fall = False
                break
            k += 1
        if fall:
            nxt = k + 1 if k < end else end
            if nxt < len(insts):
                work.append((nxt, (tuple(sorted(recent)), tuple(pending))))
    out = list(violations.values())
    if capped and verbose:
        print(f"  note: {name}: state cap reached on some blocks (analysis truncated there)", file=sys.stderr)
    return out


def extract_code_objects(lib_path, workdir):
    """gfx950 code objects of a fat binary -> list of ELF paths (a bare code object is returned as is)"""
    local = os.path.join(workdir, os.path.basename(lib_path))
    shutil.copy(lib_path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, check=False)
    objs = sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "hipv4-amdgcn" in f and "gfx950" in f)
    return objs or [local]


def disassemble(obj, workdir):
    out = obj + ".s"
    with open(out, "w") as f:
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", obj], stdout=f, stderr=subprocess.DEVNULL, check=True)
    return out


def spill_table(obj):
    """[(kernel, vgpr_spill_count, sgpr_spill_count, private bytes)] from the code object's metadata"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], capture_output=True, text=True).stdout
    rows, cur = [], {}
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            rows.append(cur)
            cur = {}
        if k in ("name", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "vgpr_count"):
            cur[k] = v
    if cur.get("name"):
        rows.append(cur)
    return [(r["name"], int(r.get("vgpr_spill_count", 0)), int(r.get("sgpr_spill_count", 0)),
             int(r.get("private_segment_fixed_size", 0))) for r in rows if "vgpr_spill_count" in r]


def lint_library(lib_path, verbose=False, spills=False):
    total_funcs = total_insts = 0
    problems = []
    with tempfile.TemporaryDirectory(prefix="vsc_lint_") as wd:
        for obj in extract_code_objects(lib_path, wd):
            funcs, base = parse_disassembly(disassemble(obj, wd))
            for name, insts in funcs.items():
                total_funcs += 1
                total_insts += len(insts)
                problems += lint_function(name, insts, base[name], verbose)
            if spills:
                for name, vs, ss, priv in spill_table(obj):
                    if vs or priv:
                        print(f"  spills: {name}: {vs} VGPRs, {ss} SGPRs, {priv} B of scratch per lane")
    return total_funcs, total_insts, problems


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("lib", nargs="?", default=os.path.join(here, "vsc2022_amd", "libvscmi.so"))
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--spills", action="store_true", help="also list the kernels that spill VGPRs (information only)")
    args = ap.parse_args()
    nf, ni, problems = lint_library(args.lib, args.verbose, args.spills)
    for p in problems:
        print(p)
    print(f"lint_isa: {nf} kernels, {ni} instructions, {len(problems)} violation(s)")
    if nf == 0:
        # (the code-object extraction fell back to the host ELF, or the library holds no gfx950 code: nothing was checked)
        print("lint_isa: NO gfx950 kernel was parsed -- that is a failed lint, not a clean one")
        return 2
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
