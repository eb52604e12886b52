"""Static scan (no GPU) for the costliest thing a register spill can do in these kernels: a scratch RELOAD that sits behind
a burst of global loads / stores.  vmcnt retires in order, so the `s_waitcnt vmcnt(0)` the compiler puts behind the reload
also waits for every older vector-memory operation — the data gradient's 16 saves + 16 loads (an HBM round trip of
microseconds) instead of the L1-resident scratch line it wanted.
Lists, per kernel of the kept assembly (libnerf_hip.so.obj/*.s), every `s_waitcnt vmcnt(N)` with N < outstanding that
follows a scratch reload while more than `--min` global operations are outstanding in program-text order.
usage: python scripts/vmcnt_drains.py [--min 4] [file.s ...]"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def drains(path, min_outstanding=4):
    text = open(path).read()
    out = []
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n", text, re.M):
        name = m.group(1)
        end = text.find(".Lfunc_end", m.end())
        if end < 0 or ".amdhsa_kernel " + name not in text:
            continue
        queue = []                     # program-text order of outstanding vector-memory ops: "g" global, "s" scratch reload, "d" LDS-DMA
        for i, line in enumerate(text[m.end():end].splitlines()):
            s = line.strip()
            if not s or s.startswith((";", ".")):
                continue
            op = s.split()[0]
            if op.startswith("scratch_load"):
                queue.append("s")
            elif op.startswith("scratch_store"):
                queue.append("t")
            elif op.startswith("global_load_lds"):
                queue.append("d")
            elif op.startswith(("global_", "buffer_", "flat_")):
                queue.append("g")
            elif op == "s_waitcnt":
                c = re.search(r"vmcnt\((\d+)\)", s)
                if not c:
                    continue
                n = int(c.group(1))
                if len(queue) > n:
                    retired = queue[:len(queue) - n]
                    if "s" in retired[-3:] and sum(x == "g" for x in retired) >= min_outstanding:
                        out.append((name, i, s, sum(x == "g" for x in retired), sum(x == "d" for x in retired)))
                    queue = queue[len(queue) - n:]
            elif op in ("s_endpgm",):
                queue = []
            elif s.endswith(":"):
                pass
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    min_out = 4
    if "--min" in args:
        k = args.index("--min")
        min_out = int(args[k + 1])
        del args[k:k + 2]
    files = args or sorted(glob.glob(os.path.join(ROOT, "nerf_amd", "csrc", "libnerf_hip.so.obj", "*.s")))
    total = 0
    for f in files:
        for name, line, wait, g, d in drains(f, min_out):
            short = re.sub(r"^_ZN?\d*_GLOBAL__N_1\d+", "", name)[:60]
            print(f"{os.path.basename(f)[:-6]:22s} {short:60s} +{line:<6d} {wait:32s} drains {g} global ops (+{d} DMA pieces) behind a scratch reload")
            total += 1
    print(f"{total} drain(s)")
