"""Generator of conv_tap6.hip's compile-time tables: the tap stream of one 64-channel group of a stride-2 3x3 half-pair convolution
(8 plane images = 4 pixel-parity planes x {hi, lo}, 27 tap slices) and the LDS-DMA schedule that keeps 4 image buffers fed.
Rules it checks: image j lives in buffer j % 4; its 10 pieces per wave may be issued from the first tap of image j - 3 (the tap after
the barrier that released the buffer) and no later than 4 taps before its own first tap (a piece of tap t has landed at the end of
tap t + 2, the hand-over barrier sits inside the last tap of image j - 1); at most SLOTS pieces per tap.
usage: python profiles/scripts/gen_tap6_tables.py   (prints the C++ arrays)"""
import itertools
import sys

SLOTS = 4
NP = 10
# (name, py, px, hl, taps [(ky, kx)], blocks)
T4 = [(0, 0), (0, 2), (2, 0), (2, 2)]
IMAGES = {
    "H11": (1, 1, 0, T4, [0, 1]), "L11": (1, 1, 1, T4, [2]),
    "H01": (0, 1, 0, [(1, 0), (1, 2)], [0, 1]), "L01": (0, 1, 1, [(1, 0), (1, 2)], [2]),
    "H10": (1, 0, 0, [(0, 1), (2, 1)], [0, 1]), "L10": (1, 0, 1, [(0, 1), (2, 1)], [2]),
    "H00": (0, 0, 0, [(1, 1)], [0, 1]), "L00": (0, 0, 1, [(1, 1)], [2]),
}


def dur(n):
    im = IMAGES[n]
    return len(im[3]) * len(im[4])


def simulate(order, groups=4):
    d = [dur(n) for n in order]
    nimg = len(order)
    start = []
    t = 0
    for g in range(groups):
        for i in range(nimg):
            start.append(t)
            t += d[i]
    total = t
    left = {j: NP for j in range(len(start))}
    # images 0, 1, 2 of group 0 come from the prologue
    for j in (0, 1, 2):
        left[j] = 0
    sched = [[] for _ in range(total)]
    for tt in range(total):
        cap = SLOTS
        for j in range(len(start)):
            if cap == 0:
                break
            if left[j] == 0:
                continue
            elig = start[j - 3] if j >= 3 else 0
            if tt < elig:
                break          # oldest first, in order
            n = min(cap, left[j])
            for k in range(n):
                sched[tt].append((j, NP - left[j] + k))
            left[j] -= n
            cap -= n
    # deadlines
    ok = True
    worst = 99
    for j in range(3, len(start) - 8):
        last = max(tt for tt in range(total) for (jj, e) in sched[tt] if jj == j)
        slack = start[j] - 4 - last
        worst = min(worst, slack)
        if slack < 0:
            ok = False
    return ok, worst, start, sched, d


best = None
names = list(IMAGES)
for perm in itertools.permutations(names[1:]):
    order = ("H11",) + perm
    ok, worst, start, sched, d = simulate(order)
    if not ok:
        continue
    # steady state must be periodic with images 0..2 of a group fully issued inside the previous group
    per = sum(d)
    g1 = [[(j - 8, e) for (j, e) in sched[per + t]] for t in range(per)]
    g2 = [[(j - 16, e) for (j, e) in sched[2 * per + t]] for t in range(per)]
    if g1 != g2:
        continue
    spill = any(j < 3 for t in range(per) for (j, e) in g1[t])   # an image 0..2 of the CURRENT group issued inside it
    if spill:
        continue
    key = (worst,)
    if best is None or key > best[0]:
        best = (key, order, g1, d)
if best is None:
    print("no valid order"); sys.exit(1)
(_, order, g1, d) = best
print("// order:", " ".join(order), " durations", d, " worst slack", best[0][0])
# per tap tables
img, ky, kx, blk = [], [], [], []
for i, n in enumerate(order):
    py, px, hl, taps, blocks = IMAGES[n]
    for b in blocks:
        for (y, x) in taps:
            img.append(i); ky.append(y); kx.append(x); blk.append(b)
assert len(img) == 27


def arr(name, v):
    print("  static constexpr int %s[%d] = {%s};" % (name, len(v), ", ".join(str(x) for x in v)))


arr("IMG", img); arr("KY", ky); arr("KX", kx); arr("BLK", blk)
arr("IM_PY", [IMAGES[n][0] for n in order]); arr("IM_PX", [IMAGES[n][1] for n in order]); arr("IM_HL", [IMAGES[n][2] for n in order])
nd = [len(g1[t]) for t in range(27)]
arr("ND", nd)
flat_i, flat_e = [], []
for t in range(27):
    row = g1[t] + [(-1, 0)] * (SLOTS - len(g1[t]))
    for (j, e) in row:
        flat_i.append(j); flat_e.append(e)
arr("DMA_IMG", flat_i)   # image index 0..7 of this group, 8..15 = of the NEXT group, -1 = none
arr("DMA_E", flat_e)
# ---- vmcnt tables.  Weight fragments: 4 loads per k-step, 8 k-steps ahead (window of 9 k-step sets).  At the end of k-step j the
# fragments of k-step j + 1 (issued at k-step j - 7) must have landed; everything issued after them may stay in flight: 7 x 4 loads
# + the pieces of k-steps j - 6 .. j (pieces are issued in the second k-step of a tap).
w0, w1 = [], []
for t in range(27):
    p3 = nd[(t - 3) % 27] + nd[(t - 2) % 27] + nd[(t - 1) % 27]
    w0.append(28 + p3)
    w1.append(28 + p3 + nd[t])
arr("WAIT0", w0); arr("WAIT1", w1)
# hand-over wait inside the last tap of image i (after group 4 of its second k-step): the pieces of image i + 1 must have landed:
# allowed in flight = operations issued after the tap that issued its last piece
start = [0]
for x in d:
    start.append(start[-1] + x)
last_piece = {}                      # image instance (8 * group + index) -> absolute tap of its last piece
for grp in range(3):
    for t in range(27):
        for (j, e) in g1[t]:
            inst = 8 * grp + j       # (entries >= 8 are images of the next group: 8 * (grp + 1) + (j - 8) = 8 * grp + j)
            last_piece[inst] = max(last_piece.get(inst, -1), 27 * grp + t)
hw = [63] * 27
for i in range(8):
    T = 27 + start[i + 1] - 1        # last tap of image i of the middle group
    tl = last_piece[8 + i + 1]
    assert tl < T, (i, tl, T)
    n = 4 * (2 * T - 2 * tl) + sum(nd[t % 27] for t in range(tl + 1, T + 1))
    hw[start[i + 1] - 1] = min(n, 63)
arr("HWAIT", hw)


# ---- perf-mode (bf16) form: one image per parity plane and 64-channel slice (no hi / lo), 9 taps per slice, plane p in buffer p
def gen_bf16():
    import itertools as it
    planes = {"P11": (1, 1, T4), "P01": (0, 1, [(1, 0), (1, 2)]), "P10": (1, 0, [(0, 1), (2, 1)]), "P00": (0, 0, [(1, 1)])}
    SL = 5                                    # pieces per tap (second k-step, groups 0..4)
    best = None
    for order in it.permutations(list(planes)):
        dd = [len(planes[n][2]) for n in order]
        st = [0]
        for x in dd:
            st.append(st[-1] + x)
        # image i of the NEXT slice may be issued from the tap after image i of this slice ends (its buffer = its plane)
        left = [NP] * 4
        sched = [[] for _ in range(9)]
        ok = True
        for t in range(9):
            cap = SL
            for i in range(4):
                if cap == 0:
                    break
                if left[i] == 0 or t < st[i + 1]:
                    continue
                n = min(cap, left[i])
                for k in range(n):
                    sched[t].append((i, NP - left[i] + k))
                left[i] -= n
                cap -= n
        if any(left):
            # the rest spills into the next slice's first taps (before the image's own first tap - 4)
            spill = [[] for _ in range(9)]
            for t in range(9):
                cap = SL - len(sched[t])
                for i in range(4):
                    if cap <= 0 or left[i] == 0:
                        continue
                    if t > st[i] - 4:
                        continue
                    n = min(cap, left[i])
                    for k in range(n):
                        spill[t].append((i + 4, NP - left[i] + k))      # + 4: image of THIS slice issued inside it
                    left[i] -= n
                    cap -= n
            if any(left):
                continue
            for t in range(9):
                sched[t] = spill[t] + sched[t]
        key = -sum(len(x) for x in sched[:2])
        if best is None or key > best[0]:
            best = (key, order, sched, dd, st)
    if best is None:
        print("// perf mode: no schedule"); return
    _, order, sched, dd, st = best
    print("// perf-mode order:", " ".join(order), "durations", dd)
    img, ky, kx = [], [], []
    for i, n in enumerate(order):
        for (y, x) in planes[n][2]:
            img.append(i); ky.append(y); kx.append(x)
    arr("B_IMG", img); arr("B_KY", ky); arr("B_KX", kx)
    arr("B_IM_PY", [planes[n][0] for n in order]); arr("B_IM_PX", [planes[n][1] for n in order])
    ndb = [len(sched[t]) for t in range(9)]
    arr("B_ND", ndb)
    fi, fe = [], []
    for t in range(9):
        row = sched[t] + [(-1, 0)] * (SL - len(sched[t]))
        for (j, e) in row:
            fi.append(j); fe.append(e)
    arr("B_DMA_IMG", fi)      # 0..3: image of the NEXT slice; 4..7: image (index - 4) of THIS slice; -1 none
    arr("B_DMA_E", fe)
    w0b, w1b = [], []
    for t in range(9):
        p3 = ndb[(t - 3) % 9] + ndb[(t - 2) % 9] + ndb[(t - 1) % 9]
        w0b.append(28 + p3); w1b.append(28 + p3 + ndb[t])
    arr("B_WAIT0", w0b); arr("B_WAIT1", w1b)
    # hand-over wait in the last tap of image i: pieces of image i + 1 (of this slice, or image 0 of the next) landed
    lastp = {}
    for grp in range(3):
        for t in range(9):
            for (j, e) in sched[t]:
                inst = 4 * (grp + 1) + j if j < 4 else 4 * grp + (j - 4)
                lastp[inst] = max(lastp.get(inst, -1), 9 * grp + t)
    hwb = [63] * 9
    for i in range(4):
        T = 9 + st[i + 1] - 1
        tl = lastp[4 + i + 1]
        assert tl < T, (i, tl, T)
        n = 4 * (2 * T - 2 * tl) + sum(ndb[t % 9] for t in range(tl + 1, T + 1))
        hwb[st[i + 1] - 1] = min(n, 63)
    arr("B_HWAIT", hwb)


gen_bf16()
