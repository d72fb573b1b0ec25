"""Static hazard check of the inline-asm load pipelines (conv_direct_kernel, conv_direct_strided_kernel).

The compiler does not know that the destination registers of an inline-asm load are written LATER.  Whatever it places
between the load and the s_waitcnt that covers it -- a register copy made by the allocator to resolve a phi, a reuse of
the register -- reads or clobbers data in flight, and the result then depends on the register allocation of the day.
This walks the control-flow graph of the device ISA of every kernel whose name matches, keeps the queue of outstanding
VMEM operations per path (loads return in order; `s_waitcnt vmcnt(N)` retires all but the newest N) and reports any
instruction that reads or writes a VGPR whose inline-asm load is still in flight; the compiler's own loads, stores and
atomics are counted for vmcnt only (the compiler waits for those itself).  Paths are explored depth-first through both
arms of every conditional branch; a block is re-entered until its entry state repeats.

Usage: python tools/check_isa.py file.s [name-pattern ...]      exit status 1 when a hazard is found
(`make -C open-universe_amd/csrc check`, also run by __graft_entry__.build())."""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def parse(lines):
    """-> (instrs, labels): instrs = [(lineno, op, args, text)], labels = {name: index into instrs}"""
    instrs, labels = [], {}
    in_asm = False
    for ln, text in lines:
        if "#ASMSTART" in text:
            in_asm = True
        elif "#ASMEND" in text:
            in_asm = False
        t = text.split(";")[0].strip()
        if not t or t.startswith("."):
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                labels[m.group(1)] = len(instrs)
            continue
        if t.endswith(":"):
            continue
        parts = t.replace(",", " ").split()
        instrs.append((ln, parts[0], parts[1:], t, in_asm))
    return instrs, labels


def check(instrs, labels, max_visits=6):
    bad = {}
    seen = {}
    stack = [(0, ())]  # (pc, pending = tuple of frozensets, oldest first)
    steps = 0
    while stack and steps < 2_000_000:
        pc, pending = stack.pop()
        pending = list(pending)
        while pc < len(instrs):
            steps += 1
            ln, op, args, t, in_asm = instrs[pc]
            if op == "s_endpgm":
                break
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    if n < len(pending):
                        pending = pending[len(pending) - n:] if n else []
                pc += 1
                continue
            if op in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz",
                      "s_cbranch_execz", "s_cbranch_execnz"):
                tgt = labels.get(args[0])
                key = (pc, tuple(pending))
                cnt = seen.get(pc, 0)
                if key in seen or cnt >= max_visits:
                    if op == "s_branch":
                        break
                else:
                    seen[key] = True
                    seen[pc] = cnt + 1
                    if tgt is not None:
                        stack.append((tgt, tuple(pending)))
                if op == "s_branch":
                    break
                pc += 1
                continue
            is_load = op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load"))
            is_store = op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "global_atomic",
                                      "buffer_atomic"))
            inflight = set().union(*pending) if pending else set()
            touched = set()
            for a in args:
                touched.update(regs(a))
            hit = touched & inflight
            if hit:
                bad[ln] = (t, sorted(hit))
            if is_load:
                # only inline-asm loads are invisible to the compiler's own wait-count insertion; its own loads are
                # covered by the waits it places (extra asm loads in the queue can only make those waits longer)
                dst = frozenset(regs(args[0])) if (in_asm and args and "lds" not in args) else frozenset()
                pending.append(dst)
            elif is_store:
                pending.append(frozenset())
            if len(pending) > 64:
                pending = pending[-64:]
            pc += 1
    return bad


def main():
    path = sys.argv[1]
    pats = sys.argv[2:] or ["conv_direct"]
    cur, kernels = None, {}
    for i, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if any(p in m.group(1) for p in pats) else None
            if cur:
                kernels[cur] = []
            continue
        if cur:
            if line.startswith(".Lfunc_end"):
                cur = None
                continue
            kernels[cur].append((i, line.rstrip()))
    total = 0
    for k, lines in kernels.items():
        instrs, labels = parse(lines)
        bad = check(instrs, labels)
        total += len(bad)
        print(f"{'ok' if not bad else str(len(bad)) + ' hazards':12s} {k}  ({len(instrs)} instructions)")
        for ln in sorted(bad)[:6]:
            print(f"     line {ln}: {bad[ln][0]}   <- in flight: v{bad[ln][1]}")
    print(f"{len(kernels)} kernels, {total} hazards")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
