"""CPU: the ISA lint (scripts/lint_isa.py) on the library that ships, and on hand-assembled snippets that hold the
hazards it exists for -- so that "0 violations" on the library means the checker looked, not that it is blind.

The hazards: a VALU-written SGPR read by a VMEM instruction less than 5 wait states
later (the `s_nop 4` of sim_i8p.hip's `bload_asm`: the compiler's hazard recognizer does not see inside inline
assembly), and a register that an outstanding, hand-waited stream load will still write being touched before its
`s_waitcnt vmcnt(N)` (what a spilled or copied ring register would look like)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
LLVM = "/opt/rocm/lib/llvm/bin"

HEAD = """
    .text
    .globl k
    .p2align 8
    .type k,@function
k:
"""
TAIL = """
    s_endpgm
.Lend:
    .size k, .Lend-k
"""
SNIPPETS = {
    # rule A: v_readfirstlane -> soffset of a buffer load
    "A_bad": ("v_readfirstlane_b32 s4, v0\n buffer_load_dword v1, v2, s[0:3], s4 offen\n s_waitcnt vmcnt(0)", {"A"}),
    "A_3_between": ("v_readfirstlane_b32 s4, v0\n s_mov_b32 s9, 0\n s_nop 1\n buffer_load_dword v1, v2, s[0:3], s4 offen\n s_waitcnt vmcnt(0)", {"A"}),
    "A_nop4": ("v_readfirstlane_b32 s4, v0\n s_nop 4\n buffer_load_dword v1, v2, s[0:3], s4 offen\n s_waitcnt vmcnt(0)", set()),
    "A_readlane_rsrc": ("v_readlane_b32 s1, v0, 3\n s_nop 2\n buffer_load_dword v1, v2, s[0:3], 0 offen\n s_waitcnt vmcnt(0)", {"A"}),
    "A_salu_write_is_fine": ("s_mov_b32 s4, 64\n buffer_load_dword v1, v2, s[0:3], s4 offen\n s_waitcnt vmcnt(0)", set()),
    "A_global_saddr": ("v_readfirstlane_b32 s6, v0\n v_readfirstlane_b32 s7, v1\n global_load_dword v3, v2, s[6:7]\n s_waitcnt vmcnt(0)", {"A"}),
    # rule B: VALU-written lane select
    "B_bad": ("v_readfirstlane_b32 s4, v0\n s_nop 1\n v_readlane_b32 s5, v1, s4", {"B"}),
    "B_ok": ("v_readfirstlane_b32 s4, v0\n s_nop 3\n v_readlane_b32 s5, v1, s4", set()),
    # rule C: m0 -> LDS-DMA
    "C_bad": ("s_mov_b32 m0, s8\n buffer_load_dword v2, s[0:3], 0 offen lds\n s_waitcnt vmcnt(0)", {"C"}),
    "C_ok": ("s_mov_b32 m0, s8\n s_nop 0\n buffer_load_dword v2, s[0:3], 0 offen lds\n s_waitcnt vmcnt(0)", set()),
    # rule D: the destination of an outstanding load
    "D_read_early": ("buffer_load_dwordx4 v[4:7], v2, s[0:3], 0 offen\n v_mov_b32 v8, v5\n s_waitcnt vmcnt(0)", {"D"}),
    "D_spill_early": ("buffer_load_dwordx4 v[4:7], v2, s[0:3], 0 offen\n scratch_store_dword off, v6, off offset:4\n s_waitcnt vmcnt(0)", {"D"}),
    "D_waited": ("buffer_load_dwordx4 v[4:7], v2, s[0:3], 0 offen\n s_waitcnt vmcnt(0)\n v_mov_b32 v8, v5", set()),
    "D_ring_partial_wait": (
        "buffer_load_dwordx4 v[4:7], v2, s[0:3], 0 offen\n buffer_load_dwordx4 v[8:11], v2, s[0:3], 0 offen offset:1024\n"
        " s_waitcnt vmcnt(1)\n v_mov_b32 v20, v4\n v_mov_b32 v21, v9\n s_waitcnt vmcnt(0)", {"D"}),
    "D_ring_partial_ok": (
        "buffer_load_dwordx4 v[4:7], v2, s[0:3], 0 offen\n buffer_load_dwordx4 v[8:11], v2, s[0:3], 0 offen offset:1024\n"
        " s_waitcnt vmcnt(1)\n v_mov_b32 v20, v4\n s_waitcnt vmcnt(0)\n v_mov_b32 v21, v9", set()),
    "D_store_counts_in_vmcnt": (   # a younger store is one more outstanding operation: vmcnt(1) still retires the load
        "buffer_load_dword v4, v2, s[0:3], 0 offen\n buffer_store_dword v9, v2, s[0:3], 0 offen\n s_waitcnt vmcnt(1)\n v_mov_b32 v20, v4", set()),
    "D_load_over_load_is_fine": ("global_load_dword v3, v9, s[6:7]\n global_load_dword v3, v9, s[6:7] offset:4\n s_waitcnt vmcnt(0)\n v_mov_b32 v1, v3", set()),
    # ... across a loop's back edge: the load of the last iteration is still outstanding behind the loop
    "D_loop_bad": (".L1:\n buffer_load_dword v4, v2, s[0:3], 0 offen\n s_add_i32 s8, s8, 1\n s_cmp_lt_i32 s8, 10\n s_cbranch_scc1 .L1\n v_mov_b32 v9, v4", {"D"}),
    "D_loop_carried_bad": (   # the ring register of iteration n is consumed in iteration n + 1 without a wait
        "buffer_load_dword v4, v2, s[0:3], 0 offen\n.L1:\n v_add_u32 v10, v10, v4\n buffer_load_dword v4, v2, s[0:3], 0 offen\n"
        " s_add_i32 s8, s8, 1\n s_cmp_lt_i32 s8, 10\n s_cbranch_scc1 .L1\n s_waitcnt vmcnt(0)", {"D"}),
    "D_loop_carried_ok": (
        "buffer_load_dword v4, v2, s[0:3], 0 offen\n.L1:\n s_waitcnt vmcnt(0)\n v_add_u32 v10, v10, v4\n buffer_load_dword v4, v2, s[0:3], 0 offen\n"
        " s_add_i32 s8, s8, 1\n s_cmp_lt_i32 s8, 10\n s_cbranch_scc1 .L1\n s_waitcnt vmcnt(0)", set()),
}


def _assemble(tmp_path, name, body):
    src = tmp_path / f"{name}.s"
    obj = tmp_path / f"{name}.o"
    src.write_text(HEAD + " " + body + TAIL)
    subprocess.run([os.path.join(LLVM, "llvm-mc"), "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-filetype=obj",
                    str(src), "-o", str(obj)], check=True, capture_output=True)
    return str(obj)


@pytest.mark.parametrize("name", sorted(SNIPPETS))
def test_lint_on_hand_assembled_snippets(tmp_path, name):
    import lint_isa

    body, want = SNIPPETS[name]
    nf, ni, problems = lint_isa.lint_library(_assemble(tmp_path, name, body))
    assert nf == 1 and ni >= 2
    rules = {p.split("[", 1)[1][0] for p in problems}
    assert rules == want, (name, problems)


def test_shipped_library_is_clean():
    """every kernel of vsc2022_amd/libvscmi.so (all 13 code objects): no VALU-SGPR -> VMEM / lane-select / m0 hazard,
    and no touch of a register that an outstanding load still owns"""
    import lint_isa

    lib = os.path.join(ROOT, "vsc2022_amd", "libvscmi.so")
    assert os.path.exists(lib), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    nf, ni, problems = lint_isa.lint_library(lib)
    assert nf >= 100 and ni > 500000, (nf, ni)   # all kernels were seen (the int8 / fp16 pre-filters alone are ~600 k instructions)
    assert not problems, "\n".join(problems[:20])
