"""Static check of the kernels that manage loads by hand -- fir_oa_kernel<NR, 16, PF>
(PF = 1, 2) of fir.hip; any kernel with tagged requests is picked up -- in the assembly
hipcc emits:

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o fir.s openseize_amd/csrc/fir.hip
    python benchmarks/check_async_regions.py fir.s

Those kernels request global loads by inline assembly (tagged `; osz:nx`, the
next pair's samples, or `; osz:hn`, its filter spectrum) and wait for them with
hand-placed `s_waitcnt vmcnt(N) ; osz:<tag>` much later; the compiler
believes the destination registers hold their values from the moment of the
request.  That is only safe while it leaves them alone: between a request and
its wait no instruction may write a destination register (a reload, a move, a
reuse as a temporary) and no spill may read one.  The check is a forward
data-flow pass over the kernel's basic blocks ("registers with a load in
flight", union at joins, to a fixed point), then one sweep that reports every
offending instruction.  `table()` returns {(nr, pf): clean?} and the reasons;
tests/test_fir_async.py holds csrc/fir_pf_table.h to it.
"""
import re
import sys

KERNEL = re.compile(r"^_ZN3osz13fir_oa_kernelILi(\d+)ELi16ELi([12])EEEvNS_7FirArgsE:")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")
READ_ONLY = ("global_store", "buffer_store", "ds_write", "s_", "v_cmp", "v_readlane",
             "v_readfirstlane", "scratch_store")


def regs(tok):
    m = REG.search(tok)
    if not m:
        return frozenset()
    if m.group(1) is not None:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    return frozenset({int(m.group(3))})


def parse(lines, nr):
    """-> blocks: list of dict(label, ins=[(lineno, kind, regs, text)], succ=[labels or 'next'])"""
    blocks = [{"label": None, "ins": [], "succ": None}]
    in_asm = False
    del nr   # the waits name what they complete; the counts are the kernel's business
    for no, line in lines:
        m = LABEL.match(line)
        if m:
            if blocks[-1]["succ"] is None:
                blocks[-1]["succ"] = ["next"]
            blocks.append({"label": m.group(1), "ins": [], "succ": None})
            continue
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        body = line.split(";")[0].strip()
        tags = frozenset(re.findall(r"osz:(\w+)", line))
        if not body or body.endswith(":") or body.startswith("."):
            continue
        ops = body.split(None, 1)
        op = ops[0]
        args = [a.strip() for a in ops[1].split(",")] if len(ops) > 1 else []
        cur = blocks[-1]
        if cur["succ"] is not None:          # code after a branch without a label: new block
            blocks.append({"label": None, "ins": [], "succ": None})
            cur = blocks[-1]
        if in_asm and op.startswith("global_load"):
            cur["ins"].append((no, "request", frozenset((r, t) for r in regs(args[0]) for t in tags), body))
        elif in_asm and op == "s_waitcnt":
            cur["ins"].append((no, "wait", tags, body))
        elif in_asm:
            continue
        elif op == "s_branch":
            cur["succ"] = [args[0]]
        elif op.startswith("s_cbranch"):
            cur["succ"] = [args[0], "next"]
        elif op == "s_endpgm":
            cur["ins"].append((no, "end", frozenset(), body))
            cur["succ"] = []
        elif op.startswith("scratch_store"):
            cur["ins"].append((no, "read", regs(args[1]) if len(args) > 1 else frozenset(), body))
        elif op.startswith(READ_ONLY):
            continue
        elif args:
            cur["ins"].append((no, "write", regs(args[0]), body))
    if blocks[-1]["succ"] is None:
        blocks[-1]["succ"] = []
    return blocks


def analyse(blocks):
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    succ = []
    for i, b in enumerate(blocks):
        out = []
        for s in b["succ"]:
            if s == "next":
                if i + 1 < len(blocks):
                    out.append(i + 1)
            elif s in index:
                out.append(index[s])
        succ.append(out)
    state_in = [frozenset() for _ in blocks]

    def transfer(i, flying, report=None):
        for no, kind, rg, text in blocks[i]["ins"]:
            if kind == "request":
                flying = flying | rg
            elif kind == "wait":
                flying = frozenset(f for f in flying if f[1] not in rg)
            elif kind == "end":
                if flying and report is not None:
                    report.append(f"line {no}: loads still in flight at s_endpgm")
            elif rg & {f[0] for f in flying} and report is not None:
                what = "spill of" if kind == "read" else "write to"
                report.append(f"line {no}: {what} a register with a load in flight: {text}")
        return flying

    work = list(range(len(blocks)))
    while work:
        i = work.pop()
        out = transfer(i, state_in[i])
        for j in succ[i]:
            merged = state_in[j] | out
            if merged != state_in[j]:
                state_in[j] = merged
                work.append(j)
    problems = []
    for i in range(len(blocks)):
        transfer(i, state_in[i], problems)
    return problems


def kernels(path):
    """{mangled kernel name: [problems]} for every kernel of the assembly file that
    requests loads by tagged inline assembly."""
    out = {}
    name, body = None, []
    for no, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append((no, line))
        if line.startswith(".Lfunc_end"):
            if any("osz:" in text and "global_load" in text for _, text in body):
                out[name] = analyse(parse(body, 0))
            name = None
    return out


def table(path):
    """fir_oa_kernel<NR, 16, PF>: {(nr, pf): clean?}, {(nr, pf): [problems]}"""
    ok, why = {}, {}
    for name, problems in kernels(path).items():
        m = KERNEL.match(name + ":")
        if m:
            key = (int(m.group(1)), int(m.group(2)))
            ok[key], why[key] = not problems, problems
    return ok, why


if __name__ == "__main__":
    for name, problems in sorted(kernels(sys.argv[1]).items()):
        print(f"{name[:60]}: {'clean' if not problems else 'NOT SAFE'}")
        for w in problems[:6]:
            print("    " + w)
    sys.exit(0)
