"""Static check of conv_tap5.hip's device assembly: the weight fragments are loaded by inline-asm global_load_dwordx4 whose data
lands asynchronously; between such a load and the second end-of-tap `s_waitcnt vmcnt(N >= 8)` after it nothing but the load itself
may touch the destination registers (a register-allocator copy / spill of them would move garbage).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include --cuda-device-only -S conv_tap5.hip -o t5.s; python check_tap5_asm.py t5.s"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
reg = re.compile(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b")


def regs_of(text):
    out = set()
    for m in reg.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


bad = 0
nload = 0
for i, ln in enumerate(lines):
    s = ln.strip()
    if not s.startswith("global_load_dwordx4 a["):
        continue
    nload += 1
    dst = regs_of(s.split(",")[0])
    waits = 0
    for j in range(i + 1, min(i + 6000, len(lines))):
        t = lines[j].strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
        if m:
            if int(m.group(1)) == 0:
                break
            if int(m.group(1)) >= 8:
                waits += 1
                if waits == 2:
                    break
            continue
        if t.startswith("s_endpgm"):
            break
        if t.startswith("v_mfma") and waits >= 1:
            continue   # (a use after the FIRST wait would be a bug of the schedule, not of the allocator: the set is read two taps later)
        if regs_of(t) & dst:
            bad += 1
            if bad <= 20:
                print("line %d: load %s touched by line %d: %s (waits passed %d)" % (i + 1, s, j + 1, t, waits))
print("weight loads: %d, violations: %d" % (nload, bad))
sys.exit(1 if bad else 0)
