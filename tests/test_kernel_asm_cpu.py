"""CPU-side guards of the register-window kernels (conv_tap5 / 6 / 6b / 7.hip): hipcc cross-compiles their device assembly here, and
(1) no inline-assembly weight load's destination registers are touched before the first MFMA that reads them (a register-allocator copy
of a fragment set whose data is still in flight would move garbage), and there is no scratch traffic (a scratch load is a full drain of
the hand-counted vmcnt queue): profiles/scripts/check_tap5_asm.py; (2) the compile-time tap-stream / DMA-schedule / vmcnt tables in
conv_tap6.hip and conv_tap6b.hip are the generator's (profiles/scripts/gen_tap6_tables.py, which checks the schedule's deadlines)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodal-learning_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("src", ["conv_tap5.hip", "conv_tap6.hip", "conv_tap6b.hip", "conv_tap7.hip"])
def test_register_window_kernels_keep_their_fragment_sets_untouched(tmp_path, src):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path / (src + ".s"))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                        "-Wno-unused-result", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    c = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "check_tap5_asm.py"), out], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout[-2000:]
    m = re.search(r"inline weight loads: (\d+), violations: 0, scratch instructions: 0", c.stdout)
    assert m and int(m.group(1)) >= 100, c.stdout[-500:]


def _arrays(text):
    return {m.group(1): [int(v) for v in m.group(2).replace("\n", " ").split(",") if v.strip()]
            for m in re.finditer(r"static constexpr int (\w+)\[\d+\] = \{([^}]*)\};", text)}


def test_tap6_tables_are_the_generators():
    g = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "scripts", "gen_tap6_tables.py")], capture_output=True, text=True)
    assert g.returncode == 0, g.stderr[-2000:]
    gen = _arrays(g.stdout)
    hp = _arrays(open(os.path.join(CSRC, "conv_tap6.hip")).read())
    for k in ("IMG", "KY", "KX", "BLK", "IM_PY", "IM_PX", "IM_HL", "ND", "DMA_IMG", "DMA_E", "WAIT0", "WAIT1", "HWAIT"):
        assert hp[k] == gen[k], k
    pm = _arrays(open(os.path.join(CSRC, "conv_tap6b.hip")).read())
    for k in ("IMG", "KY", "KX", "IM_PY", "IM_PX", "ND", "DMA_IMG", "DMA_E", "WAIT0", "WAIT1", "HWAIT"):
        assert pm[k] == gen["B_" + k], k
