"""CPU tests of the lane plumbing behind the twin launches (ops.Pair, ops.lanewise, unet._Twin): pure host logic -- the
kernels themselves are covered by tests/test_kernels.py::test_twin_* (emulation + MI355X) and the networks by
tests/test_models.py::test_twin_trunk_equals_the_two_network_evaluation (-m gpu)."""
import torch

from editanything_amd import ops
from editanything_amd.unet import _Twin, _twin_value


def test_pair_behaves_like_two_tensors():
    a, b = torch.arange(24.0).view(2, 3, 4), torch.arange(24.0).view(2, 3, 4) + 100
    p = ops.Pair(a, b)
    assert p.shape == a.shape and p.dtype == a.dtype and p.device == a.device
    v = p.view(6, 4)
    assert torch.equal(v.a, a.view(6, 4)) and torch.equal(v.b, b.view(6, 4))
    s = p[..., :2]
    assert torch.equal(s.a, a[..., :2]) and torch.equal(s.b, b[..., :2])
    d = ops.dup_rows(p)
    assert d.a.shape[0] == 4 and torch.equal(d.b[:2], b) and torch.equal(d.b[2:], b)
    assert torch.equal(ops.dup_rows(a, dim=1), torch.cat([a, a], 1))
    c = ops.cols(ops.Pair(a.view(6, 4), b.view(6, 4)), ops.Pair(0, 2), 2)
    assert torch.equal(c.a, a.view(6, 4)[:, 0:2]) and torch.equal(c.b, b.view(6, 4)[:, 2:4])
    assert torch.equal(ops.cols(a.view(6, 4), 1, 2), a.view(6, 4)[:, 1:3])
    half = ops.Pair(None, b)                       # an optional operand only one lane has
    assert half[0].a is None and torch.equal(half[0].b, b[0])


def test_lane_and_zip_round_trip_through_nested_values():
    a, b = torch.ones(2), torch.zeros(2)
    val = (ops.Pair(a, b), 3, [ops.Pair(b, a), None], {"k": ops.Pair(1, 2)})
    assert ops._has_pair(val) and not ops._has_pair((a, 3, [None]))
    l0, l1 = ops._lane(val, 0), ops._lane(val, 1)
    assert l0[0] is a and l1[0] is b and l0[1] == 3 and l0[2][0] is b and l1[2][0] is a and l0[3]["k"] == 1 and l1[3]["k"] == 2
    z = ops._zip((a, 7, None, (b, 2)), (b, 7, None, (a, 2)))
    assert isinstance(z[0], ops.Pair) and z[1] == 7 and z[2] is None and isinstance(z[3][0], ops.Pair) and z[3][1] == 2
    n0, n1 = ops.Normed(a, a, 1e-5, True), ops.Normed(b, b, 1e-5, True)
    zn = ops._zip(n0, n1)
    assert isinstance(zn, ops.Pair) and zn.a is n0 and zn.b is n1        # per-lane objects stay per lane


def test_lanewise_runs_once_per_lane_and_zips_the_results():
    calls = []

    @ops.lanewise
    def f(x, scale=1.0, extra=None):
        calls.append(float(x.sum()))
        return x * scale, (x.shape[0], None)
    a, b = torch.ones(3), torch.full((3,), 2.0)
    out, meta = f(a, scale=2.0)
    assert len(calls) == 1 and torch.equal(out, a * 2) and meta == (3, None)
    out, meta = f(ops.Pair(a, b), scale=ops.Pair(2.0, 3.0), extra=(ops.Pair(a, b), 1))
    assert len(calls) == 3 and torch.equal(out.a, a * 2) and torch.equal(out.b, b * 3) and meta == (3, None)


def test_twin_module_proxy_pairs_tensors_and_passes_equal_scalars():
    class M:
        pass
    ma, mb = M(), M()
    ma.w, mb.w = torch.ones(2, 2), torch.zeros(2, 2)
    ma.ln, mb.ln = [(torch.ones(2), torch.zeros(2))], [(torch.zeros(2), torch.ones(2))]
    ma.cout = mb.cout = 320
    ma.emb_off, mb.emb_off = 0, 640
    ma.skip_w = mb.skip_w = None
    t = _Twin(ma, mb)
    assert isinstance(t.w, ops.Pair) and t.w.a is ma.w and t.w.b is mb.w
    assert t.cout == 320 and t.skip_w is None
    assert isinstance(t.emb_off, ops.Pair) and (t.emb_off.a, t.emb_off.b) == (0, 640)
    assert isinstance(t.ln, list) and isinstance(t.ln[0], tuple) and isinstance(t.ln[0][0], ops.Pair) and t.ln[0][1].b is mb.ln[0][1]
    assert _twin_value(3, 3) == 3


def test_fusion_switches_are_per_thread_and_per_object():
    """ops.current(): the process default outside every block, an object's own copy inside `ops.using(cfg)` (what
    ControlledDenoiser(fusion=...) wraps its evaluations in), nested blocks restore, other threads are unaffected, unknown
    switches raise."""
    import threading
    import pytest
    assert ops.current() is ops.CONFIG and ops.CONFIG.gn_next is True
    mine = ops.make_config(gn_next=False)
    assert mine.gn_next is False and mine.ln_fold == ops.CONFIG.ln_fold and ops.CONFIG.gn_next is True
    seen = {}
    with ops.using(mine):
        assert ops.current() is mine
        with ops.using(None):                       # an object without its own switches: whatever is in force stays
            assert ops.current() is mine
        with ops.using(ops.make_config(ln_fold=False)) as inner:
            assert ops.current() is inner and inner.ln_fold is False
        assert ops.current() is mine
        th = threading.Thread(target=lambda: seen.setdefault("other", ops.current()))
        th.start()
        th.join()
    assert seen["other"] is ops.CONFIG and ops.current() is ops.CONFIG
    with pytest.raises(TypeError):
        ops.make_config(no_such_switch=True)
