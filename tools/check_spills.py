"""Build gate on the generated device code (`make check`, run by __graft_entry__.build()).

1. No kernel may use scratch or spill VGPRs: every `private_segment_fixed_size` / `vgpr_spill_count` in the code-object
   metadata must be 0.
2. The step loop of the GRU ring kernels must not spill SGPRs to VGPR lanes.  Rare-path diagnostics inlined into it once cost
   10 SGPRs -> `v_writelane_b32` / `v_readlane_b32` pairs in the loop -> +25 us per 401-frame pass.  What is counted: every
   `v_writelane_b32` that lies inside a loop of a `gru_ring_kernel` instantiation (between a label and a backward branch to
   it) and is NOT part of a call sequence -- the save / restore pairs the compiler places right around an `s_swappc_b64`
   (the out-of-line `gru_stale_probe` / `gru_note_long_wait` calls in the rare poll branch) are legitimate and are recognised
   by their shape: the next read of the same (register, lane) follows a call, both within CALL_WINDOW instructions of it.

Usage: python tools/check_spills.py file.s [file.s ...]      exit status 1 when a gate fails."""
import re
import sys

CALL_WINDOW = 48
LOOP_KERNELS = ("gru_ring_kernel",)
# Kernels whose 256 accumulation registers fill the AGPR file: the allocator parks a few dozen accumulator values in scratch ONCE,
# between the main loop and the epilogue (conv_splitw_kernel: 27 dwords per lane and launch).  Allowed there -- and only
# there: a scratch access inside a loop of such a kernel fails the gate like any other spill.
SCRATCH_OUTSIDE_LOOPS_OK = ("conv_splitw_kernel",)


def kernels_of(lines):
    """-> {name: [instruction / label lines]} for every function of the file.  A function starts at its label line
    (`_Zname:` optionally followed by `; @_Zname`) and ends at `.Lfunc_end`."""
    out, cur = {}, None
    for line in lines:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        out[cur].append(line.rstrip("\n"))
    return out


def loop_spills(body):
    """v_writelane_b32 instructions inside a loop and away from any call -> list of (index, text)."""
    instrs = []  # (text, label or None)
    labels = {}
    for line in body:
        t = line.split(";")[0].strip()
        if not t or (t.startswith(".") and not re.match(r"^\.LBB\w+:", t)):
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if t.endswith(":"):
            continue
        instrs.append(t)
    in_loop = [False] * len(instrs)
    for i, t in enumerate(instrs):
        parts = t.replace(",", " ").split()
        if parts[0].startswith(("s_cbranch", "s_branch")) and len(parts) > 1:
            tgt = labels.get(parts[1])
            if tgt is not None and tgt <= i:  # backward branch: [tgt, i] is a loop body
                for k in range(tgt, i + 1):
                    in_loop[k] = True
    calls = [i for i, t in enumerate(instrs) if t.startswith(("s_swappc_b64", "s_setpc_b64"))]

    def lane_of(t):  # "v_writelane_b32 v90, s76, 0" / "v_readlane_b32 s76, v90, 0" -> ("v90", "0")
        a = [x.strip() for x in t.split(None, 1)[1].split(",")]
        return (a[0], a[2]) if t.startswith("v_writelane") else (a[1], a[2])

    bad = []
    for i, t in enumerate(instrs):
        if not t.startswith("v_writelane_b32") or not in_loop[i]:
            continue
        # a call save: the next read of the same (register, lane) comes after a call, and both sit right around that call
        key = lane_of(t)
        rd = next((k for k in range(i + 1, len(instrs)) if instrs[k].startswith("v_readlane_b32") and lane_of(instrs[k]) == key),
                  None)
        call = next((c for c in calls if c > i), None)
        if rd is not None and call is not None and call < rd and call - i <= CALL_WINDOW and rd - call <= CALL_WINDOW:
            continue
        bad.append((i, t))
    return bad


def loop_scratch(body):
    """scratch_* instructions inside a loop -> list of texts."""
    instrs, labels = [], {}
    for line in body:
        t = line.split(";")[0].strip()
        if not t or (t.startswith(".") and not re.match(r"^\.LBB\w+:", t)):
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if t.endswith(":"):
            continue
        instrs.append(t)
    in_loop = [False] * len(instrs)
    for i, t in enumerate(instrs):
        parts = t.replace(",", " ").split()
        if parts[0].startswith(("s_cbranch", "s_branch")) and len(parts) > 1:
            tgt = labels.get(parts[1])
            if tgt is not None and tgt <= i:
                for k in range(tgt, i + 1):
                    in_loop[k] = True
    return [t for i, t in enumerate(instrs) if in_loop[i] and t.startswith(("scratch_", "buffer_store_dword v", "buffer_load_dword v")) and "scratch" in t]


def check_file(path):
    lines = open(path).read().split("\n")
    fails = []
    cur = ""
    for ln, line in enumerate(lines, 1):
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            cur = m.group(1)
        if re.search(r"(private_segment_fixed_size|vgpr_spill_count):\s+[1-9]", line):
            if any(k in cur for k in SCRATCH_OUTSIDE_LOOPS_OK):
                continue  # (held to "nothing inside a loop" below)
            fails.append(f"{path}:{ln}: scratch / VGPR spills in the device code: {line.strip()} ({cur})")
    for name, body in kernels_of(lines).items():
        if any(k in name for k in SCRATCH_OUTSIDE_LOOPS_OK):
            bad = loop_scratch(body)
            if bad:
                fails.append(f"{path}: {name} accesses scratch inside a loop ({len(bad)} instructions, first: {bad[0]})")
    n_loop_kernels = 0
    for name, body in kernels_of(lines).items():
        if not any(k in name for k in LOOP_KERNELS):
            continue
        n_loop_kernels += 1
        bad = loop_spills(body)
        if bad:
            fails.append(f"{path}: {name} spills SGPRs inside a loop ({len(bad)} v_writelane_b32, first: {bad[0][1]})")
    return fails, n_loop_kernels


def main(argv):
    fails, n = [], 0
    for path in argv:
        f, k = check_file(path)
        fails += f
        n += k
    for f in fails:
        print(f)
    if fails:
        return 1
    print(f"no scratch, no VGPR spills in {len(argv)} file(s); {n} GRU ring kernels: no SGPR spills inside their loops")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
