"""CPU: the spill gate of `make check` (tools/check_spills.py) on hand-written ISA.  Round 3's gate was an awk range whose
pattern never matched the real label lines (`name: ; @name`), so it passed whatever the compiler did; this one must FAIL on
an SGPR spill inside a loop of a gru_ring_kernel instantiation and on scratch / VGPR spills anywhere, and must accept the
save / restore pairs around an out-of-line call."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_spills", os.path.join(ROOT, "tools", "check_spills.py"))
C = importlib.util.module_from_spec(spec)
spec.loader.exec_module(C)

NAME = "_ZN2ou15gru_ring_kernelILi4ELi8ELb1EEEvNS_7GruArgsEi"
HEAD = f"\t.globl\t{NAME}\n{NAME}: ; @{NAME}\n; %bb.0:\n\ts_load_dwordx2 s[0:1], s[4:5], 0x0\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n\t.size x, .Lfunc_end0-x\n"
META = "    .private_segment_fixed_size: 0\n    .sgpr_spill_count: 0\n    .vgpr_spill_count: 0\n"


def run(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return C.check_file(str(p))


def test_label_with_comment_suffix_is_recognised(tmp_path):
    fails, n = run(tmp_path, HEAD + TAIL + META)
    assert n == 1 and not fails


def test_sgpr_spill_inside_the_step_loop_fails(tmp_path):
    loop = (".LBB0_1: ; =>This Loop Header\n\tv_writelane_b32 v90, s44, 0\n\tv_add_f32_e32 v1, v2, v3\n"
            "\tv_readlane_b32 s44, v90, 0\n\ts_cbranch_scc1 .LBB0_1\n")
    fails, n = run(tmp_path, HEAD + loop + TAIL + META)
    assert n == 1 and len(fails) == 1 and "spills SGPRs inside a loop" in fails[0]
    assert C.main([str(tmp_path / "k.s")]) == 1


def test_call_save_around_an_out_of_line_function_is_accepted(tmp_path):
    loop = (".LBB0_1:\n\tv_writelane_b32 v90, s76, 0\n\tv_writelane_b32 v90, s77, 1\n\ts_getpc_b64 s[4:5]\n"
            "\ts_swappc_b64 s[30:31], s[4:5]\n\tv_readlane_b32 s76, v90, 0\n\tv_readlane_b32 s77, v90, 1\n"
            "\ts_cbranch_scc1 .LBB0_1\n")
    fails, _ = run(tmp_path, HEAD + loop + TAIL + META)
    assert not fails


def test_spill_outside_any_loop_is_not_the_gates_business(tmp_path):
    body = "\tv_writelane_b32 v90, s44, 0\n\tv_readlane_b32 s44, v90, 0\n"
    fails, _ = run(tmp_path, HEAD + body + TAIL + META)
    assert not fails


def test_a_spill_that_is_read_before_the_call_is_a_spill(tmp_path):
    # same loop as the call save, but the lane is read back BEFORE the call: not a save around it
    loop = (".LBB0_1:\n\tv_writelane_b32 v90, s76, 0\n\tv_readlane_b32 s76, v90, 0\n\ts_swappc_b64 s[30:31], s[4:5]\n"
            "\ts_cbranch_scc1 .LBB0_1\n")
    fails, _ = run(tmp_path, HEAD + loop + TAIL + META)
    assert len(fails) == 1


def test_scratch_and_vgpr_spills_fail_for_any_kernel(tmp_path):
    other = "_ZN2ou11fir4_kernelILi9EEEvPKfS2_f: ; @x\n\ts_endpgm\n.Lfunc_end1:\n"
    for meta in ("    .private_segment_fixed_size: 32\n", "    .vgpr_spill_count: 3\n"):
        fails, _ = run(tmp_path, other + meta)
        assert len(fails) == 1 and "scratch / VGPR spills" in fails[0]
    fails, _ = run(tmp_path, other + META)
    assert not fails


def test_other_kernels_may_spill_sgprs(tmp_path):
    # (conv_chain_kernel does, in its set-up code; the loop rule is specific to the recurrence kernels)
    other = ("_ZN2ou17conv_chain_kernelILi1ELi8EEEvNS_9ChainArgsEii: ; @x\n.LBB1_1:\n\tv_writelane_b32 v9, s4, 0\n"
             "\ts_cbranch_scc1 .LBB1_1\n\ts_endpgm\n.Lfunc_end1:\n")
    fails, n = run(tmp_path, other + META)
    assert not fails and n == 0
