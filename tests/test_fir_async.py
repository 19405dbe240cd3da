"""The FIR kernel variants that keep loads in flight in registers the compiler
does not know about (csrc/fir_pair.h: FirPair::nx, FirPair::Hn) are only
dispatched where the emitted assembly leaves those registers alone: rebuild
the assembly of fir.hip with the same flags as the library and hold
csrc/fir_pf_table.h to benchmarks/check_async_regions.py.  (CPU test: hipcc
cross-compiles gfx950 without a GPU.)"""

import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openseize_amd", "csrc")


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


@pytest.mark.skipif(_hipcc() is None, reason="needs hipcc")
def test_dispatched_fir_variants_keep_inflight_registers_untouched(tmp_path):
    asm = tmp_path / "fir.s"
    subprocess.run([_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                    "-o", str(asm), os.path.join(CSRC, "fir.hip")], check=True, cwd=CSRC,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    try:
        import check_async_regions as chk
    finally:
        sys.path.pop(0)
    ok, why = chk.table(str(asm))
    assert sorted(ok) == [(nr, pf) for nr in range(8, 16) for pf in (1, 2)], sorted(ok)
    text = open(os.path.join(CSRC, "fir_pf_table.h")).read()
    m = re.search(r"#define OSZ_FIR_PF_TABLE \{([^}]*)\}", text)
    chosen = [int(v) for v in m.group(1).split(",")]
    assert len(chosen) == 8 and all(v in (0, 1, 2) for v in chosen)
    for nr, pf in zip(range(8, 16), chosen):
        if pf:
            assert ok[(nr, pf)], (nr, pf, why[(nr, pf)][:4])


def test_checker_sees_a_planted_violation(tmp_path):
    """The data-flow pass is not vacuous: a write to a register with a load in
    flight, reached only through a branch, is reported; the same write after the
    wait is not."""
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    try:
        import check_async_regions as chk
    finally:
        sys.path.pop(0)
    head = "_ZN3osz13fir_oa_kernelILi8ELi16ELi1EEEvNS_7FirArgsE:\n"
    load = "\t;;#ASMSTART\n\tglobal_load_dwordx2 v[10:11], v[2:3], off ; osz:nx\n\t;;#ASMEND\n"
    wait = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(16) ; osz:nx\n\t;;#ASMEND\n"
    tail = "\ts_endpgm\n.Lfunc_end0:\n"
    bad = head + load + "\ts_cbranch_vccnz .LBB0_2\n\tv_mov_b32_e32 v1, 0\n.LBB0_2:\n\tv_add_f64 v[10:11], v[4:5], v[6:7]\n" + wait + tail
    good = head + load + "\ts_cbranch_vccnz .LBB0_2\n\tv_mov_b32_e32 v1, 0\n.LBB0_2:\n" + wait + "\tv_add_f64 v[10:11], v[4:5], v[6:7]\n" + tail
    spill = head + load + "\tscratch_store_dwordx2 off, v[10:11], off\n" + wait + tail
    for name, text, clean in (("bad", bad, False), ("good", good, True), ("spill", spill, False)):
        f = tmp_path / f"{name}.s"
        f.write_text(text)
        ok, why = chk.table(str(f))
        assert ok == {(8, 1): clean}, (name, ok, why)
