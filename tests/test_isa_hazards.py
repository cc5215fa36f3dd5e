"""CPU: the hand-written VMEM stores of persist.hip / exact.hip are hazard-free IN THE EMITTED ISA.

The on-chip CG kernel and the assembly write through inline-asm `global_store_dwordx4 ... sc1` stores.  LLVM's hazard
recognizer neither looks at the uses inside inline asm nor knows what the asm's last instruction needs from the code
behind it, so two gfx9/CDNA hazards are the source's job (kernels.h, MAG_WS_SBASE / MAG_WS_DATA):

  H1  VALU writes an SGPR (a spilled SGPR restored by v_readlane_b32) -> VMEM reads it as its base: 5 wait states.
      Cause of round 3's GPU faults of the stamped k_cg_persist build (v_readlane_b32 s89 two instructions before
      global_store_dwordx4 ..., s[88:89]).
  H2  store of more than 8 bytes -> next instructions rewrite its data registers: 2 wait states (commit 78fa2a0: K
      wrong at 1M triangles only, because only a backed-up memory pipeline reads the data late enough to see it).

Both are timing- and allocation-dependent on the GPU (H2 needed a 1M-triangle mesh to show, H1 a particular register
allocation), which is why this check runs on the ISA text instead: `make isa` emits it with the product's and the
diagnostic build's exact flags, scripts/isa_lint.py scans every inline-asm store.  The same sources compiled with
-DMAG_ASM_NO_WAITSTATES (the asm's own s_nops dropped) MUST be objected to -- including the two historical instances,
as long as the compiler still produces them."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "magnetite_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="needs hipcc to emit the ISA")


@pytest.fixture(scope="module")
def isa():
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC, "isa", "isa-nows"], stdout=subprocess.DEVNULL)
    b = os.path.join(CSRC, "build")
    return {n: os.path.join(b, n + ".s") for n in ("persist", "persist_k4", "persist_stamps", "exact", "persist_nows",
                                                    "persist_k4_nows", "persist_stamps_nows", "exact_nows")}


# (persist_k4: the translation unit that holds the four-slot edge-block instantiation alone, under the max-ilp scheduler)
@pytest.mark.parametrize("name,min_stores", [("persist", 100), ("persist_k4", 8), ("persist_stamps", 100), ("exact", 8)])
def test_emitted_isa_has_no_store_hazard(isa, name, min_stores):
    import isa_lint
    assert isa_lint.count_asm_stores(isa[name]) >= min_stores  # the scan saw the stores it is about
    problems = isa_lint.lint(isa[name])
    assert problems == [], "\n".join(problems[:20])


@pytest.mark.parametrize("name", ["persist_nows", "persist_k4_nows", "persist_stamps_nows", "exact_nows"])
def test_lint_objects_when_the_wait_states_are_dropped(isa, name):
    import isa_lint
    problems = isa_lint.lint(isa[name])
    h2s = [p for p in problems if "H2 structural" in p]
    assert len(h2s) >= isa_lint.count_asm_stores(isa[name]) // 2  # every asm block with a wide store (two stores per block at most)
    if name.startswith("persist"):
        assert any("H1 structural" in p for p in problems)  # put_granules_at / put_granules_sys_at: SGPR base


def test_lint_parses_the_two_historical_hazards():
    """The two instances that bit, as text: round 3's stale-base store and commit 78fa2a0's rewritten data registers."""
    import isa_lint
    import tempfile
    h1 = """
	v_readlane_b32 s88, v254, 12
	v_readlane_b32 s89, v254, 13
	v_mov_b32_e32 v4, v3
	;;#ASMSTART
	global_store_dwordx4 v4, v[12:15], s[88:89] sc1
	global_store_dwordx4 v4, v[16:19], s[88:89] offset:16 sc1
	s_nop 1
	;;#ASMEND
	s_nop 0
	s_nop 0
"""
    h2 = """
	;;#ASMSTART
	global_store_dwordx4 v[4:5], v[0:3], off sc1
	;;#ASMEND
	v_add_f64 v[10:11], v[8:9], v[6:7]
	v_cndmask_b32_e64 v0, -1, v36, s[8:9]
	s_nop 0
"""
    ok = """
	v_readlane_b32 s89, v254, 13
	;;#ASMSTART
	s_nop 4
	global_store_dwordx4 v4, v[12:15], s[88:89] sc1
	global_store_dwordx4 v4, v[16:19], s[88:89] offset:16 sc1
	s_nop 1
	;;#ASMEND
	v_mov_b32_e32 v12, 0
"""
    with tempfile.TemporaryDirectory() as d:
        for text, want in ((h1, "H1: `v_readlane_b32 s89"), (h2, "H2: `v_cndmask_b32_e64 v0")):
            p = os.path.join(d, "t.s")
            open(p, "w").write(text)
            got = isa_lint.lint(p)
            assert any(want in g for g in got), got
        p = os.path.join(d, "ok.s")
        open(p, "w").write(ok)
        assert isa_lint.lint(p) == []
