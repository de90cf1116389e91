#!/usr/bin/env python3
"""Instruction budget of one kernel from the compiler's assembly (hipcc -S --cuda-device-only ... srx_trace.hip).
Splits the kernel into basic blocks (labels and fall-through after branches), counts VALU / SALU / LDS / VMEM /
branch / wait instructions per block (asm blocks between ;;#ASMSTART / ;;#ASMEND included, s_nop counted apart), and
prints one line per block plus, with --sections file, the sums over named groups of blocks.
Usage: isa_budget.py trace.s <mangled kernel name> [--sections sections.txt]   (sections.txt: `name | LBB1_41 LBB1_46 ...`)"""
import re, sys
from collections import OrderedDict, defaultdict

def classify(op):
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_call")): return "branch"
    if op.startswith(("s_setprio", "s_barrier", "s_sleep")): return "other"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("v_"): return "valu"
    return "other"

def main():
    path, kernel = sys.argv[1], sys.argv[2]
    sections = None
    if "--sections" in sys.argv:
        sections = OrderedDict()
        for line in open(sys.argv[sys.argv.index("--sections") + 1]):
            if "|" in line and not line.startswith("#"):
                n, b = line.rsplit("|", 1); sections[n.strip()] = b.split()
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks = OrderedDict(); cur = "entry"; blocks[cur] = defaultdict(int)
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m: cur = m.group(1)[1:]; blocks.setdefault(cur, defaultdict(int)); continue
        m = re.match(r"^; %bb\.(\d+):", s)
        if m: cur = "bb." + m.group(1); blocks.setdefault(cur, defaultdict(int)); continue
        if not s or s.startswith((";", ".", "//")): continue
        blocks[cur][classify(s.split()[0])] += 1
    kinds = ["valu", "salu", "lds", "vmem", "smem", "branch", "wait", "nop", "other"]
    print("block".ljust(12) + "".join(k.rjust(8) for k in kinds))
    for b, c in blocks.items():
        print(b.ljust(12) + "".join(str(c[k]).rjust(8) for k in kinds))
    if sections:
        print("\n" + "section".ljust(62) + "".join(k.rjust(8) for k in kinds))
        seen = set()
        for n, bl in sections.items():
            tot = defaultdict(int)
            for b in bl:
                if b not in blocks: print(f"  (unknown block {b})"); continue
                seen.add(b)
                for k in kinds: tot[k] += blocks[b][k]
            print(n.ljust(62) + "".join(str(tot[k]).rjust(8) for k in kinds))
        rest = [b for b in blocks if b not in seen]
        print("blocks in no section: " + " ".join(rest))

if __name__ == "__main__":
    main()
