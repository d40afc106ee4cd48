"""CPU: the LDS bank model behind the sample-ring layouts (tools/lds_bank_model.py).

The model applies the lane groups / bank functions MI355X_MICROARCH.md gives per DS instruction to the addresses the
kernel's window reads and ring writes generate.  These checks pin the design claims: the split layout + phi is
conflict-free for the bit-transposed lane map of the two-wave N = 2048 frames at EVERY window offset (reads and writes),
the plain layout is conflict-free for the natural map at even offsets, and the plain layout under the transposed map
costs what the PMC counters measured before the split layout existed (profiles/r03_generic_pmc.txt)."""
import importlib.util
import os

from conftest import ROOT


def _model():
    spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(ROOT, "tools", "lds_bank_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_split_layout_is_conflict_free_for_the_transposed_lane_map():
    m = _model()
    n = 2048
    c, lay = m.layouts(n)["split4+phi"]
    assert c == n + 256 and (c // 4) % 64 == 0          # a region is a multiple of the 64 indices phi permutes
    floor = (n // 16 // 64) * 8 * 2 * 2                  # 2 waves x 8 rows x 2 dwords x 2 cycles per ds_read_b32
    cycles = [m.window_cycles(n, m.j0_xlm, c, lay, p) for p in range(c)]
    assert set(cycles) == {floor}                         # every offset, odd or even
    t = n // 16
    for w0 in range(0, c, 4):                             # ring writes: 4 ds_write_b32 per wave and row, 4 cycles each
        assert sum(m.wr_b32([lay((w0 + 4 * (64 * wv + l)) % c + r) for l in range(64)]) for wv in range(t // 64) for r in range(4)) == 32


def test_plain_layout_costs_what_was_measured():
    m = _model()
    # natural lane map (N = 1024): one conflict-free ds_read_b64 per point at even offsets; two 2-way conflicting dword
    # reads per point at odd ones
    c, lay = m.layouts(1024)["inter"]
    ev = [m.window_cycles(1024, m.j0_natural, c, lay, p) for p in range(0, c, 2)]
    od = [m.window_cycles(1024, m.j0_natural, c, lay, p) for p in range(1, c, 2)]
    assert max(ev) <= 17 and min(ev) >= 16 and 64 <= min(od) and max(od) <= 66
    # transposed lane map on the plain layout: 4-way (even) / 8-way (odd) conflicts -- the 452 conflict cycles per frame
    # rocprofv3 counted for the first version of the N = 2048 ring kernel (mean of the two parities at an odd hop: ~322
    # over the 32-cycle floor, plus the conflicting writes)
    c, lay = m.layouts(2048)["inter"]
    ev = m.window_cycles(2048, m.j0_xlm, c, lay, 0)
    od = m.window_cycles(2048, m.j0_xlm, c, lay, 1)
    assert 128 <= ev <= 132 and 512 <= od <= 518
    # without phi the split layout still leaves a 2-way conflict
    c, lay = m.layouts(2048)["split4"]
    assert m.window_cycles(2048, m.j0_xlm, c, lay, 7) == 128
