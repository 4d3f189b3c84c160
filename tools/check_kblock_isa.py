#!/usr/bin/env python
"""Static check of the instruction streams of kernel_block_pp, kernel_block_one and kernel_block_res (manifold_gp_amd/csrc/features.hip;
kernel_block_res: see check_res) and of spmm_mt_kernel (csrc/spmm.hip: see check_replay).

Its staging loads are inline asm (`buffer_load_dwordx4 ... offen` through descriptors of the operands' exact extents) whose completion the compiler does not track: the kernel waits for
them with its own `s_waitcnt vmcnt(0)`.  That is only sound if NO instruction touches a destination register of such a load
between the load and the next vmcnt wait.  This script compiles features.hip to gfx950 assembly and walks every
kernel_block_pp<TS> instantiation in text order, which is execution order here: between a load and its wait the kernel has no
branch (the check fails if it finds one); a label there is where the path that skipped the loads joins.  It also checks that the tile's 16
stores are the only VMEM stores, that each is a 16-byte buffer store, and that nothing writes their data registers before
kb_store_done's s_nop (the compiler's own hazard rule does not cover the register-soffset form, see features.hip).
Exit code 0 = clean.  Used by tests/test_host_cpu.py; needs hipcc, no GPU."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def check(asm):
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN\d+_GLOBAL__N_1\d+kernel_block_(pp|one)ILi\d+EE.*:\s*(;.*)?$", l)]
    problems, seen = [], 0
    for a in starts:
        b = next(i for i in range(a, len(lines)) if ".end_amdhsa_kernel" in lines[i])
        name = lines[a].split(":")[0]
        seen += 1
        inflight, stores, store_regs, after_stores = set(), 0, set(), False
        for l in lines[a + 1:b]:
            t = l.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue        # a label inside the window is a join with a path that issued no load: the window stays open
            op = t.split()[0]
            m = re.match(r"buffer_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\], s\d+ offen", t)
            if m:
                inflight |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                continue
            if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
                problems.append("%s: a load the check does not know: %s" % (name, t))
            if op == "s_waitcnt" and "vmcnt(0)" in t:
                inflight = set()
                continue
            if inflight:
                if op.startswith("s_cbranch") or op == "s_branch":
                    problems.append("%s: branch between a staging load and its wait: %s" % (name, t))
                if regs_of(t) & inflight:
                    problems.append("%s: touches an in-flight staging register: %s" % (name, t))
            if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("flat_store"):
                if op != "buffer_store_dwordx4":
                    problems.append("%s: unexpected store %s" % (name, t))
                stores += 1
                store_regs |= regs_of(t.split(",")[0])
                after_stores = True
                continue
            if after_stores:
                if op == "s_nop":
                    after_stores, store_regs = False, set()
                elif op.startswith("v_") and regs_of(t.split(",")[0]) & store_regs:
                    problems.append("%s: writes a store's data register before the s_nop: %s" % (name, t))
        if stores != 16:
            problems.append("%s: %d stores, expected the tile's 16" % (name, stores))
    if seen != 8:
        problems.append("expected 4 instantiations each of kernel_block_pp and kernel_block_one, found %d in all" % seen)
    return problems


def check_res(asm, expect=29):
    """kernel_block_res<Q>: the streamed operand's loads are inline asm and the kernel waits for them with `s_waitcnt vmcnt(N)`,
    N > 0, counting on the wave's vector memory operations completing in order.  For each of its two loops (the walk with two
    row blocks and the one with one) this replays the loop body three times against an in-order queue of its loads and stores:
    entered with nothing in flight (a `vmcnt(0)` and no memory operation between it and the loop head), a `vmcnt(N)` retires all
    but the N youngest operations, and no instruction may read or write a destination register of a load still in the queue.
    Also: the body is straight-line, holds ceil(Q / 2) loads and 16 stores per row block, every store a dword store."""
    lines = asm.split("\n")
    starts = [(i, int(m.group(1))) for i, l in enumerate(lines)
              for m in [re.match(r"^_ZN\d+_GLOBAL__N_1\d+kernel_block_resILi(\d+)EE.*:\s*(;.*)?$", l)] if m]
    problems = []
    for a, q in starts:
        b = next(i for i in range(a, len(lines)) if ".end_amdhsa_kernel" in lines[i])
        name = "kernel_block_res<%d>" % q
        body = [(i, lines[i].strip()) for i in range(a + 1, b)]
        body = [(i, t) for i, t in body if t and not t.startswith(";") and not t.startswith(".") or re.match(r"^\.LBB\d+_\d+:", t)]
        labels = {t.split(":")[0]: k for k, (i, t) in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", t)}
        loops = []
        for k, (i, t) in enumerate(body):
            m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", t)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((labels[m.group(1)], k))
        if len(loops) != 2:
            problems.append("%s: %d loops, expected the two walks" % (name, len(loops)))
            continue
        for head, tail in loops:
            # entry: back from the loop head to the nearest vmcnt(0); nothing in between may be a memory operation or a label
            k = head - 1
            while k >= 0 and not (body[k][1].startswith("s_waitcnt") and "vmcnt(0)" in body[k][1]):
                t = body[k][1]
                if t.startswith((".LBB", "buffer_", "global_", "flat_", "scratch_")):
                    problems.append("%s: between the prologue's vmcnt(0) and the loop head: %s" % (name, t))
                k -= 1
            if k < 0:
                problems.append("%s: no vmcnt(0) in front of a loop" % name)
            inner = [t for i, t in body[head + 1:tail]]
            if any(t.startswith(".LBB") or t.startswith("s_cbranch") or t.startswith("s_branch") for t in inner):
                problems.append("%s: the loop body is not straight-line" % name)
            nload = sum(t.startswith("buffer_load_dwordx4") for t in inner)
            nstore = sum(t.startswith("buffer_store") for t in inner)
            if nload != (q + 1) // 2 or nstore not in (16, 32) or any(t.startswith("buffer_store") and not t.startswith("buffer_store_dword v") for t in inner):
                problems.append("%s: %d loads / %d stores in a loop body (expected %d / 16 or 32 dword stores)" % (name, nload, nstore, (q + 1) // 2))
            queue = []          # in issue order: a set of destination registers per load, None per store
            for it in range(3):
                for t in inner:
                    op = t.split()[0]
                    m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
                    if m:
                        n = int(m.group(1))
                        if len(queue) > n:
                            queue = queue[len(queue) - n:]
                        continue
                    inflight = set().union(*[r for r in queue if r]) if queue else set()
                    if regs_of(t) & inflight:
                        problems.append("%s: touches a register of a load still in flight (iteration %d): %s" % (name, it, t))
                    if op.startswith("buffer_load"):
                        m = re.match(r"buffer_load_dwordx4 v\[(\d+):(\d+)\]", t)
                        if not m:
                            problems.append("%s: a load the check does not know: %s" % (name, t))
                            continue
                        queue.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
                    elif op.startswith("buffer_store"):
                        queue.append(None)
                    elif op.startswith(("global_", "flat_", "scratch_")):
                        problems.append("%s: unexpected memory instruction: %s" % (name, t))
    if len(starts) != expect:
        problems.append("expected %d instantiations of kernel_block_res, found %d" % (expect, len(starts)))
    return problems


def check_replay(asm, pattern, expect, what):
    """Kernels whose loads are inline asm waited for with `vmcnt(N)`, N > 0, and whose loop is ENTERED with loads in flight
    (spmm_mt_kernel: three blocks of operands ahead): the whole function is replayed against an in-order queue of its vector
    memory operations.  The stream is walked as one path: unconditional forward branches are followed, every loop (a backward
    branch, conditional or not; loops side by side, not nested; a loop the compiler rotated, i.e. entered in its middle, too) runs
    three times and is left through its exit.  Every loop body is replayed on TWO paths: with every conditional forward branch
    that stays inside the body falling through (the optional code -- spmm_mt_kernel's tile boundary and pending epilogue -- runs in
    every round: its loads must not be touched before the waits have retired them) and with every such branch taken (no optional
    operation in the queue: the counted waits then retire the fewest operations, the case they are counted for).  No instruction
    may read or write a destination register of a load still in the queue; at s_endpgm the queue must be empty of loads."""
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(pattern, l)]
    problems = []
    br = re.compile(r"s_c?branch\w* (\.LBB\d+_\d+)")
    ubr = re.compile(r"s_branch (\.LBB\d+_\d+)")
    cbr = re.compile(r"s_cbranch_\w+ (\.LBB\d+_\d+)")
    for a in starts:
        b = next(i for i in range(a, len(lines)) if ".end_amdhsa_kernel" in lines[i])
        name = what + " " + re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", lines[a].split(":")[0])[:40]
        body = [lines[i].strip() for i in range(a + 1, b)]
        body = [t for t in body if t and not t.startswith(";") and (not t.startswith(".") or re.match(r"^\.LBB\d+_\d+:", t))]
        labels = {t.split(":")[0]: k for k, t in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", t)}
        back = {}
        for k, t in enumerate(body):
            m = br.match(t)
            if m and m.group(1) in labels and labels[m.group(1)] <= k:
                back[labels[m.group(1)]] = k
        if not back or any(h1 < h2 <= t1 for h1, t1 in back.items() for h2 in back if h2 != h1):
            problems.append("%s: %d loops, nested or none" % (name, len(back)))
            continue

        def linear(lo, hi, take):
            out, k = [], lo
            while k < hi:
                t = body[k]
                out.append(t)
                mu, mc = ubr.match(t), cbr.match(t)
                if mu and labels.get(mu.group(1), -1) > k:
                    k = min(labels[mu.group(1)], hi)
                elif take and mc and k < labels.get(mc.group(1), -1) < hi:
                    k = labels[mc.group(1)]
                else:
                    k += 1
            return out

        def leave(head, tail):
            if ubr.match(body[tail]):       # behind an unconditional back edge: the body's last conditional branch that leaves it
                exits = [labels[m.group(1)] for m in (cbr.match(body[q]) for q in range(head, tail)) if m and labels.get(m.group(1), -1) > tail]
                return exits[-1] if exits else tail + 1
            return tail + 1

        for take in (False, True):
            trace, k = [], 0
            while k < len(body):
                if k in back:
                    trace += linear(k, back[k] + 1, take) * 3
                    k = leave(k, back[k])
                    continue
                t = body[k]
                trace.append(t)
                mu, mb = ubr.match(t), br.match(t)
                if mu and labels.get(mu.group(1), -1) > k:
                    k = labels[mu.group(1)]
                elif mb and mb.group(1) in labels and labels[mb.group(1)] <= k and back.get(labels[mb.group(1)]) == k:
                    head = labels[mb.group(1)]          # the back edge of a loop entered in its middle: two more rounds from its head
                    trace += linear(head, k + 1, take) * 2
                    k = leave(head, k)
                else:
                    k += 1
            queue, mfma = [], 0
            for t in trace:
                op = t.split()[0]
                if op.startswith(".LBB"):
                    continue
                m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    if len(queue) > n:
                        queue = queue[len(queue) - n:]
                    continue
                if op == "s_endpgm":
                    if any(r for r in queue):
                        problems.append("%s: loads in flight at s_endpgm" % name)
                    queue = []
                    continue
                inflight = set().union(*[r for r in queue if r]) if queue else set()
                if regs_of(t) & inflight:
                    problems.append("%s: touches a register of a load still in flight: %s" % (name, t))
                mfma += op.startswith("v_mfma")
                # (flat_load: the compiler's own loads of optional epilogue operands, which it waits for with vmcnt(0) lgkmcnt(0) before
                # their first use: in the queue like the others, so that a touch of their destinations in flight is seen too)
                m = re.match(r"(buffer|global|flat)_load_dword(x(\d))? v(\[(\d+):(\d+)\]|(\d+))", t)
                if m:
                    lo = int(m.group(5) or m.group(7)); hi = int(m.group(6) or m.group(7))
                    queue.append(set(range(lo, hi + 1)))
                elif op.startswith(("buffer_load", "global_load", "flat_load", "scratch_")):
                    problems.append("%s: a load the check does not know: %s" % (name, t))
                elif op.startswith(("buffer_store", "global_store", "flat_store", "global_atomic", "buffer_atomic")):
                    queue.append(None)
            if mfma == 0:
                problems.append("%s: no MFMA found" % name)
    if len(starts) != expect:
        problems.append("expected %d instantiations of %s, found %d" % (expect, what, len(starts)))
    return problems


def compile_to_asm(src, extra=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "out.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.dirname(src), "--cuda-device-only", "-S", src, "-o", out] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            return None
        return open(out).read()


MT_PATTERN = r"^_ZN\d+_GLOBAL__N_1\d+spmm_mt_kernelILb[01]EE.*:\s*(;.*)?$"


def main():
    csrc = os.path.join(ROOT, "manifold_gp_amd", "csrc")
    asm = compile_to_asm(os.path.join(csrc, "features.hip"))
    if asm is None:
        return 2
    problems = check(asm) + check_res(asm)
    # spmm.hip is built with the MFMA accumulators in VGPRs (csrc/build.sh)
    asm = compile_to_asm(os.path.join(csrc, "spmm.hip"), ["-mllvm", "-amdgpu-mfma-vgpr-form=1"])
    if asm is None:
        return 2
    problems += check_replay(asm, MT_PATTERN, 2, "spmm_mt_kernel")
    for p in problems:
        print(p)
    print("kernel_block_pp / kernel_block_one / kernel_block_res / spmm_mt_kernel instruction streams:", "clean" if not problems else "%d problem(s)" % len(problems))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
