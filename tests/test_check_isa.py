"""CPU: the static hazard checker of the inline-asm load pipelines (tools/check_isa.py, a build gate) on hand-written ISA:
it must flag a register touched while its asm load is in flight -- on the straight path and around a loop back-edge --
and stay quiet once the counted wait has retired the load, and for the compiler's own loads."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
C = importlib.util.module_from_spec(spec)
spec.loader.exec_module(C)


def run(text):
    lines = list(enumerate(text.strip().splitlines(), 1))
    instrs, labels = C.parse(lines)
    return C.check(instrs, labels)


ASM_LOAD = """
	;;#ASMSTART
	buffer_load_dword v10, v1, s[4:7], s2 offen
	;;#ASMEND
"""


def test_copy_of_a_register_in_flight_is_flagged():
    bad = run(ASM_LOAD + "\tv_mov_b32_e32 v20, v10\n\ts_waitcnt vmcnt(0)\n\ts_endpgm")
    assert len(bad) == 1 and "v_mov_b32_e32 v20, v10" in list(bad.values())[0][0]


def test_use_after_the_counted_wait_is_fine():
    ok = run(ASM_LOAD + ASM_LOAD.replace("v10", "v11") +
             "\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v20, v10\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v21, v11\n\ts_endpgm")
    assert not ok
    bad = run(ASM_LOAD + ASM_LOAD.replace("v10", "v11") + "\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v21, v11\n\ts_endpgm")
    assert len(bad) == 1  # the second load is still outstanding (loads return in order)


def test_hazard_around_a_loop_back_edge():
    # the load issued at the bottom of the loop is read by the copy at the top of the next iteration
    text = ".LBB0_1:\n\tv_mov_b32_e32 v30, v10\n\ts_waitcnt vmcnt(0)\n" + ASM_LOAD + "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm"
    bad = run(text)
    assert len(bad) == 1


def test_compiler_loads_and_stores_only_count_for_vmcnt():
    # a compiler-generated load (outside ASMSTART / ASMEND) is the compiler's business; but it does occupy a vmcnt slot
    text = ("\tglobal_load_dword v10, v[2:3], off\n\tv_mov_b32_e32 v20, v10\n" + ASM_LOAD.replace("v10", "v11") +
            "\tglobal_store_dword v[2:3], v5, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v21, v11\n\ts_endpgm")
    assert not run(text)
    text = ASM_LOAD.replace("v10", "v11") + "\tglobal_store_dword v[2:3], v5, off\n\ts_waitcnt vmcnt(2)\n\tv_mov_b32_e32 v21, v11\n\ts_endpgm"
    assert len(run(text)) == 1
