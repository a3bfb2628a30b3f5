"""GPU tests (``-m gpu``) of the FUSED step of precision mode 'bf16x3' on [hi | lo] pair planes (include/morgana_hip.h, "The FUSED step
of precision mode 'bf16x3' on PAIR PLANES"; functional.F0StackX3Fn): every entry point through the C ABI against float64 arithmetic on
the operands it was given, the whole step against fp32 mode, the generic 'bf16x3' path and the numpy oracle, the captured step against
the eager loop.  Reference arithmetic is fp32 (morgana/experiment_builder.py:262-263, morgana/data.py:127); the bar is the north
star's 1e-4 of the largest element, the observed errors (a few 1e-6) are recorded by conftest's parity report."""
import numpy as np
import pytest
import torch

from morgana_amd import _lib, data, graphs, models, ops, optim, synthetic, utils
from morgana_amd import functional as F_hip
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'
RTOL = 1e-4


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t if dtype is None else t.to(dtype)


from parity_report import rel_err          # noqa: E402,F401  max |got - want| / max |want|, recorded per test (gpurun_out/parity_report.json)


def pair_value(pair, cols):
    """float64 (rows, cols) value hi + lo of a pair-plane buffer."""
    ldp = pair.shape[1] // 2
    return pair[:, :cols].double() + pair[:, ldp:ldp + cols].double()


def _load_state(model, state):
    own = model.state_dict()
    for key, value in state.items():
        own[key].copy_(torch.from_numpy(value))
    return model


@pytest.mark.parametrize('rows,cols,extra', [(37, 600, 0), (256, 128, 5), (1024, 512, 0), (5, 9, 3)])
def test_split_pair_planes(rows, cols, extra):
    """mg_split3_bf16 order 5: hi = bf16(x), lo = bf16(x - hi) side by side in one row, zero padding columns, zero extra rows, plain and
    transposed - bit for bit the same arithmetic in torch - and hi + lo carries x to 2^-16."""
    rng = np.random.RandomState(rows + cols)
    x = dev((rng.standard_normal((rows, cols)) * np.exp(rng.uniform(-3, 3, (rows, 1)))).astype(np.float32))
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    p = ops.split_pair(x, extra_rows=extra)
    ldp = ops.pad_ld(cols)
    assert tuple(p.shape) == (rows + extra, 2 * ldp)
    assert torch.equal(p[:rows, :cols], hi) and torch.equal(p[:rows, ldp:ldp + cols], lo)
    assert not bool(p[:, cols:ldp].any()) and not bool(p[:, ldp + cols:].any()) and not bool(p[rows:].any())
    assert float((pair_value(p[:rows], cols) - x.double()).abs().max() / x.abs().max()) < 2.0 ** -16
    t = ops.split_pair(x, transpose=True)
    ldt = ops.pad_ld(rows)
    assert tuple(t.shape) == (cols, 2 * ldt)
    assert torch.equal(t[:, :rows], hi.t()) and torch.equal(t[:, ldt:ldt + rows], lo.t())
    assert not bool(t[:, rows:ldt].any()) and not bool(t[:, ldt + rows:].any())


@pytest.mark.parametrize('m,k,n,act', [(21504, 600, 512, ops.ACT_SIGMOID), (4500, 600, 512, ops.ACT_SIGMOID), (6144, 512, 256, ops.ACT_NONE),
                                       (70000, 600, 512, ops.ACT_SIGMOID)])
def test_forward_gemm_on_pair_planes(m, k, n, act):
    """mg_phone_front_linear_fwd_x3 (the GEMM alone): the output pair's value against float64 arithmetic on the pair operands' values -
    what is dropped is the lo x lo product (2^-16 relative per product) and fp32 accumulation - and the pair is a proper split."""
    rng = np.random.RandomState(m % 1000 + k)
    x = dev(rng.rand(m, k).astype(np.float32))
    w = dev((rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    bias = dev((rng.standard_normal(n) * 0.1).astype(np.float32))
    a2, w2 = ops.split_pair(x), ops.split_pair(w)
    (y2,) = ops.phone_front_x3(None, (a2, k, w2, bias, n, act))
    assert tuple(y2.shape) == (m, 2 * n)
    want = pair_value(a2, k) @ pair_value(w2, k).t() + bias.double()
    if act == ops.ACT_SIGMOID:
        want = torch.sigmoid(want)
    got = pair_value(y2, n)
    assert float((got - want).abs().max() / want.abs().max()) < 2e-5
    hi = y2[:, :n].float()
    lo = y2[:, n:].float()
    assert float((lo.abs() / hi.abs().clamp_min(1e-30)).max()) <= 2.0 ** -8           # lo is the rounding remainder of hi
    for tune in (89,):                                                                  # A/B: three passes over the plane - the same products
        ops_lib = _lib.load()
        _lib.check(ops_lib.mg_set_tuning(7, tune), 'mg_set_tuning')
        try:
            (y3,) = ops.phone_front_x3(None, (a2, k, w2, bias, n, act))
        finally:
            _lib.check(ops_lib.mg_set_tuning(7, 0), 'mg_set_tuning')
        assert float((pair_value(y3, n) - want).abs().max() / want.abs().max()) < 2e-5


@pytest.mark.parametrize('parts', [1, 3])
def test_f32_forward_gemm_on_pair_planes(parts):
    """mg_linear_fwd_x3_f32, the 512 -> 128 layer: fp32 output (one buffer, or the three products' partial sums) against float64."""
    rng = np.random.RandomState(3)
    m, k, n = 21504, 512, 128
    x = dev(rng.rand(m, k).astype(np.float32))
    w = dev((rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    bias = dev((rng.standard_normal(n) * 0.1).astype(np.float32))
    a2, w2 = ops.split_pair(x), ops.split_pair(w)
    y = ops.linear_fwd_x3_f32(a2, k, w2, bias, n, ops.ACT_NONE, parts=parts)
    got = y.double().sum(0) if parts == 3 else y.double()
    want = pair_value(a2, k) @ pair_value(w2, k).t() + bias.double()
    assert float((got - want).abs().max() / want.abs().max()) < 1e-5


def test_tail_rows_x3_equals_the_f32_tail():
    """mg_f0_tail_rows_x3 against mg_f0_tail_rows_f32 on the same pre-activations: prediction and the tail's gradients the same numbers
    (the same fma chains), dZ2 the [hi | lo] split of the fp32 dZ2, db2 its column sums; three partial pre-activation buffers add up."""
    rng = np.random.RandomState(5)
    m = 21504
    z2 = dev(rng.standard_normal((m, 128)).astype(np.float32))
    w3 = dev((rng.standard_normal((32, 128)) * 0.1).astype(np.float32))
    b3 = dev((rng.standard_normal(32) * 0.1).astype(np.float32))
    w4 = dev((rng.standard_normal((1, 32)) * 0.2).astype(np.float32))
    b4 = dev(np.array([0.1], np.float32))
    ybar = dev(rng.standard_normal(m).astype(np.float32))
    weight = dev((rng.rand(m) / m).astype(np.float32))
    pred0, dz0, flat0 = ops.f0_tail_rows_f32(z2, w3, b3, w4, b4, ybar, weight)
    n = ops.F0_TAIL_X3_N
    for parts in (1, 3):
        if parts == 3:
            z_in = torch.stack((z2 * 0.5, z2 * 0.25, z2 * 0.25))           # exact binary fractions: the sum is z2 bit for bit
        else:
            z_in = z2
        pred, dz2p, ws, n_slabs, stride = ops.f0_tail_rows_x3(z_in, w3, b3, w4, b4, ybar, weight)
        flat = ops.slab_reduce(ws, n_slabs, stride, ops.F0_TAIL_X3_SLAB, torch.empty(ops.F0_TAIL_X3_SLAB, device=DEV))
        assert torch.equal(pred, pred0)
        assert torch.equal(flat[128:n], flat0[:n - 128])                  # dW3 | db3 | dW4 | db4 | loss
        hi = dz0.to(torch.bfloat16)
        lo = (dz0 - hi.float()).to(torch.bfloat16)
        assert torch.equal(dz2p[:, :128], hi) and torch.equal(dz2p[:, 128:], lo)
        want_db2 = dz0.double().sum(0)
        assert float((flat[:128].double() - want_db2).abs().max() / dz0.abs().double().sum(0).max()) < 1e-6


@pytest.mark.parametrize('m', [21504, 5000, 30000, 96])
def test_l2tail_x3_equals_the_two_launches(m):
    """mg_f0_l2tail_x3 (the 512 -> 128 layer on pair planes + the exact-fp32 tail, Z2 on chip) against mg_linear_fwd_x3_f32 +
    mg_f0_tail_rows_x3: the layer's products in another order of sums (1e-6 on the prediction), the tail the same fma chains; row counts
    that take one round of workgroups, a ragged last block, several blocks per workgroup and a single block."""
    rng = np.random.RandomState(m % 977)
    h1 = dev(rng.rand(m, 512).astype(np.float32) * 0.8 + 0.1)
    w2 = dev((rng.standard_normal((128, 512)) / np.sqrt(512)).astype(np.float32))
    b2 = dev((rng.standard_normal(128) * 0.1).astype(np.float32))
    w3 = dev((rng.standard_normal((32, 128)) * 0.1).astype(np.float32))
    b3 = dev((rng.standard_normal(32) * 0.1).astype(np.float32))
    w4 = dev((rng.standard_normal((1, 32)) * 0.2).astype(np.float32))
    b4 = dev(np.array([0.1], np.float32))
    ybar = dev(rng.standard_normal(m).astype(np.float32))
    weight = dev((rng.rand(m) / m).astype(np.float32))
    h1p, w2p = ops.split_pair(h1), ops.split_pair(w2)
    z2 = ops.linear_fwd_x3_f32(h1p, 512, w2p, b2, 128, ops.ACT_NONE)
    pred0, dz0, ws0, s0, st0 = ops.f0_tail_rows_x3(z2, w3, b3, w4, b4, ybar, weight)
    flat0 = ops.slab_reduce(ws0, s0, st0, ops.F0_TAIL_X3_SLAB, torch.empty(ops.F0_TAIL_X3_SLAB, device=DEV)).clone()
    pred0, dz0 = pred0.clone(), dz0.clone()
    pred, dz, ws, s, st = ops.f0_l2tail_x3(h1p, w2p, b2, w3, b3, w4, b4, ybar, weight)
    flat = ops.slab_reduce(ws, s, st, ops.F0_TAIL_X3_SLAB, torch.empty(ops.F0_TAIL_X3_SLAB, device=DEV))
    assert float((pred - pred0).abs().max() / pred0.abs().max()) < 2e-6
    n = ops.F0_TAIL_X3_N
    for lo_, hi_ in ((0, 128), (128, 128 + 4096), (128 + 4096, n - 1)):
        assert float((flat[lo_:hi_] - flat0[lo_:hi_]).abs().max() / flat0[lo_:hi_].abs().max().clamp_min(1e-30)) < 2e-5
    assert float((flat[n - 1] - flat0[n - 1]).abs() / flat0[n - 1].abs()) < 2e-6
    got, want = pair_value(dz, 128), pair_value(dz0, 128)
    assert float((got - want).abs().max() / want.abs().max()) < 2e-5


def test_pair_plane_backward_kernels():
    """mg_linear_wgrad_dgrad_x3 (dW2 slabs, dZ1 pair, the column sums of dZ1) and mg_linear_wgrad_slabs_x3 (dW1 slabs; one walk with all
    four planes per stage, and three walks: MG_TUNE_AB 90) against float64 arithmetic on the pair operands' values."""
    rng = np.random.RandomState(9)
    m, n2, k1, k0 = 21504, 128, 512, 600
    h1 = dev(rng.rand(m, k1).astype(np.float32) * 0.8 + 0.1)
    dz2 = dev((rng.standard_normal((m, n2)) * 1e-3).astype(np.float32))
    w2 = dev((rng.standard_normal((n2, k1)) / np.sqrt(k1)).astype(np.float32))
    x = dev(rng.rand(m, k0).astype(np.float32))
    h1p, dz2p, w2tp, xp = ops.split_pair(h1), ops.split_pair(dz2), ops.split_pair(w2, transpose=True), ops.split_pair(x)
    slab2, s2, st2, dz1p, colsum, n_cs = ops.linear_wgrad_dgrad_x3(dz2p, h1p, m, n2, k1, w2tp)
    hv, gv = pair_value(h1p, k1), pair_value(dz2p, n2)
    dw2 = ops.slab_reduce(slab2, s2, st2, n2 * k1, torch.empty(n2 * k1, device=DEV)).view(n2, k1).double()
    want_dw2 = gv.t() @ hv
    assert float((dw2 - want_dw2).abs().max() / want_dw2.abs().max()) < 1e-5
    want_dz1 = (gv @ pair_value(w2tp, n2).t()) * hv * (1.0 - hv)
    got_dz1 = pair_value(dz1p, k1)
    assert float((got_dz1 - want_dz1).abs().max() / want_dz1.abs().max()) < 2e-5
    db1 = ops.slab_reduce(colsum, n_cs, k1, k1, torch.empty(k1, device=DEV)).double()
    assert float((db1 - want_dz1.sum(0)).abs().max() / want_dz1.abs().sum(0).max()) < 1e-5
    want_dw1 = got_dz1.t() @ pair_value(xp, k0)
    lib = _lib.load()
    for tune in (0, 90):
        _lib.check(lib.mg_set_tuning(7, tune), 'mg_set_tuning')
        try:
            slab1, s1, st1 = ops.linear_wgrad_slabs_x3(dz1p, xp, m, k1, k0)
        finally:
            _lib.check(lib.mg_set_tuning(7, 0), 'mg_set_tuning')
        dw1 = ops.slab_reduce(slab1, s1, st1, k1 * k0, torch.empty(k1 * k0, device=DEV)).view(k1, k0).double()
        assert float((dw1 - want_dw1).abs().max() / want_dw1.abs().max()) < 1e-5, tune


def _step_results(precision, fused, feats_np, ragged=False):
    utils.X3_FUSED = fused
    try:
        model = _load_state(models.F0Model(precision=precision).to(DEV), synthetic.f0_model_state())
        feats = data.to_device(feats_np, DEV, bf16_tables=model.bf16_table_features())
        _lib.CALL_LOG = []
        loss, out = model(feats)
        loss.backward()
        calls = list(_lib.CALL_LOG)
    finally:
        _lib.CALL_LOG = None
        utils.X3_FUSED = True
    return (loss.item(), out['pred_norm_lf0'].detach().cpu().numpy(), {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()},
            calls)


@pytest.mark.parametrize('b,t,ragged', [(64, 1000, False), (96, 700, True)])
def test_fused_x3_step_against_fp32_mode_and_the_generic_path(b, t, ragged):
    """functional.F0StackX3Fn (plain autograd hand-over) against fp32 mode and the generic 'bf16x3' path on batches the fused form takes:
    loss, prediction and every gradient to 1e-4 of the largest element - the bar of fp32 mode - fixed and ragged lengths (padding
    frames: the table's extra rows carry their bias gradients)."""
    feats_np = synthetic.make_batch(b, t, seed=21)
    if ragged:
        rng = np.random.RandomState(4)
        n_frames = feats_np['n_frames'].copy()
        dur = feats_np['dur'].copy()
        for i in range(b):                                   # shorten every other utterance: drop trailing phones
            if i % 2:
                keep = rng.randint(dur.shape[1] // 2, dur.shape[1])
                dur[i, keep:] = 0
                n_frames[i] = int(dur[i].sum())
        feats_np = dict(feats_np, dur=dur, n_frames=n_frames)
    want = _step_results('fp32', True, feats_np)
    generic = _step_results('bf16x3', False, feats_np)
    fused = _step_results('bf16x3', True, feats_np)
    assert 'mg_phone_front_linear_fwd_x3' in fused[3] and 'mg_phone_front_linear_fwd_x3' not in generic[3]
    for got in (generic, fused):
        np.testing.assert_allclose(got[0], want[0], rtol=RTOL)
        assert rel_err(got[1], want[1]) < RTOL
        for name in want[2]:
            assert rel_err(got[2][name], want[2][name]) < RTOL, name


def test_fused_x3_step_at_c2_against_the_oracle():
    """BASELINE config C2 (256 x 1000 frames) through the fused 'bf16x3' step against the numpy oracle: loss and prediction to 1e-4,
    every gradient to 1e-4 of its largest element."""
    feats = synthetic.make_batch(256, 1000)
    state = synthetic.f0_model_state()
    want_loss, want_pred, want_grads = ref_cpu.f0_forward_backward(state, feats)
    got = _step_results('bf16x3', True, feats)
    assert 'mg_phone_front_linear_fwd_x3' in got[3]
    np.testing.assert_allclose(got[0], want_loss, rtol=RTOL)
    assert rel_err(got[1], want_pred) < RTOL
    for name in want_grads:
        assert rel_err(got[2][name], want_grads[name]) < RTOL, name


def test_update_kernel_keeps_the_weight_pairs_current_and_graph_equals_eager():
    """The captured step (graphs.GraphedTrainStep: the update kernel re-splits the weights it changed, the forward's tail is finished by
    the update launch) against the eager loop (optim.Adam with fused_loop) over the same eight steps: parameters and the last loss bit
    for bit, the weights' pair planes equal to a fresh split of the weights, and FIVE entry-point calls = five launches per captured step
    (<= 8: VERDICT round 4, item 1)."""
    feats_np = synthetic.make_batch(64, 1000, seed=5)
    results = {}
    for mode in ('eager', 'graph'):
        model = _load_state(models.F0Model(precision='bf16x3').to(DEV), synthetic.f0_model_state())
        feats = data.to_device(feats_np, DEV, bf16_tables=model.bf16_table_features())
        assert 'normalised_lab' + data.X3_TABLE_SUFFIX in feats
        opt = optim.Adam(model.parameters(), lr=0.01, fused_loop=True)
        if mode == 'eager':
            for _ in range(8):
                opt.zero_grad()
                loss, _ = model(feats)
                F_hip.backward(loss)
                opt.step()
            last = loss.item()
        else:
            _lib.CALL_LOG = []
            try:
                step = graphs.GraphedTrainStep(model, opt, feats, warmup=3)          # three eager steps, then the capture (runs nothing)
                log = list(_lib.CALL_LOG)
            finally:
                _lib.CALL_LOG = None
            eager_calls = len(log) // 4 if False else None
            captured = log[len(log) - 6:]
            captured = log[len(log) - 5:]
            assert captured == ['mg_phone_front_linear_fwd_x3', 'mg_f0_l2tail_x3', 'mg_linear_wgrad_dgrad_x3', 'mg_linear_wgrad_slabs_x3',
                                'mg_adam_step_plan_f32'], log[-10:]
            for _ in range(5):
                step()
            last = step.loss.item()
        torch.cuda.synchronize()
        for lin, want_t in ((model.layers[0], False), (model.layers[2], True)):
            pr = lin.weight._mg_pair
            assert torch.equal(pr['plain'], ops.split_pair(lin.weight.detach()))
            if want_t:
                assert torch.equal(pr['t'], ops.split_pair(lin.weight.detach(), transpose=True))
        results[mode] = ({n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()}, last)
    assert results['eager'][1] == results['graph'][1]
    for name in results['eager'][0]:
        assert np.array_equal(results['eager'][0][name], results['graph'][0][name]), name
