"""Static check of the device assembly of conv_tap5.hip / conv_tap6.hip: the weight fragments are loaded by inline-asm
global_load_dwordx4 (recognisable by the s_mov_b64 of the base in front) whose data lands asynchronously, long after the compiler
considers the destination defined.  Between such a load and the FIRST MFMA that reads the registers (by then the schedule's waits
have covered it) nothing may touch them: a register-allocator copy / spill of a set in flight would move garbage.  Also reports scratch traffic (a
scratch load is a full drain of the hand-counted vmcnt queue).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include --cuda-device-only -S conv_tap5.hip -o t5.s; python check_tap5_asm.py t5.s"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")


def regs(text, kind):
    out = set()
    for m in re.finditer(r"\b%s\[(\d+):(\d+)\]|\b%s(\d+)\b" % (kind, kind), text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def is_inline_load(i):
    t = lines[i].strip()
    if not re.match(r"global_load_dwordx4 [av]\[", t):
        return False
    k = i - 1
    while k > 0 and (not lines[k].strip() or lines[k].strip().startswith(";")):
        k -= 1
    # conv_tap5/6: s_mov_b64 of the scalar base in front; conv_tap7: vector address, `off` - the only dwordx4 loads with a register
    # destination in these kernels (the compiler's own are buffer loads)
    return ((lines[k].strip().startswith("s_mov_b64") or lines[k].strip().startswith("s_nop")) and ", off" not in t) or (", off" in t and "offset:" in t)


bad = 0
n = 0
for i, ln in enumerate(lines):
    if not is_inline_load(i):
        continue
    t = ln.strip()
    m = re.match(r"global_load_dwordx4 ([av])\[(\d+):(\d+)\]", t)
    kind = m.group(1)
    dst = set(range(int(m.group(2)), int(m.group(3)) + 1))
    n += 1
    for j in range(i + 1, min(i + 40000, len(lines))):
        u = lines[j].strip()
        if not u or u.startswith(";") or u.startswith("."):
            continue
        if u.startswith("s_endpgm") or u.startswith("s_waitcnt vmcnt(0)") or u.startswith("s_branch") or u.startswith("s_setpc"):
            break                       # (a full drain: whatever touches the registers afterwards sees landed data; an unconditional
                                        # branch: the text that follows is not this path - the scan is linear)
        if not (regs(u, kind) & dst):
            continue
        if u.startswith("v_mfma"):
            break                       # first use: the in-flight window is over
        bad += 1
        if bad <= 20:
            print("line %d: %s touched by line %d: %s" % (i + 1, t, j + 1, u))
        break
scratch = sum(1 for l in lines if l.strip().startswith("scratch_"))
print("inline weight loads: %d, violations: %d, scratch instructions: %d" % (n, bad, scratch))
sys.exit(1 if (bad or scratch) else 0)
