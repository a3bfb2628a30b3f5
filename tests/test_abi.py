"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports exactly what include/morgana_hip.h
declares, the ctypes signature table covers it, and the product path refuses to run without a device / library."""
import os
import re

import numpy as np
import pytest
import torch

import morgana_amd
from morgana_amd import _lib, ops

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, 'include', 'morgana_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mg_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
    declared = _declared_symbols()
    assert len(declared) >= 25
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), 'libmorgana_hip.so does not export %s' % name
    assert sorted(_lib.SIGNATURES) == declared


def test_library_identity_calls():
    lib = _lib.load()
    assert lib.mg_version() == 1
    assert lib.mg_build_arch() == b'gfx950'
    assert lib.mg_masked_mse_workspace_bytes(4, 100, 3) >= 4 * 4
    assert lib.mg_linear_wgrad_workspace_bytes(256000, 512, 600) >= 512 * 600 * 4
    assert lib.mg_gru_bwd_workspace_bytes(64, 512) >= 64 * 512 * 4


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.load()
    rc = lib.mg_upsample_index(None, 0, 0, 0, None, None, None)
    assert rc == -1 and 'mg_upsample_index' in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, 'mg_upsample_index')


def test_philox_block_function_known_answers():
    """mg_philox4x32_10 (host side of csrc/dropout.hip: the generator mg_dropout draws its masks from) against the three known-answer
    vectors of Random123's Philox4x32-10 (Salmon et al., SC'11; kat_vectors of the Random123 distribution)."""
    import ctypes
    lib = _lib.load()

    def block(counter, key):
        c, k, out = (ctypes.c_uint32 * 4)(*counter), (ctypes.c_uint32 * 2)(*key), (ctypes.c_uint32 * 4)()
        lib.mg_philox4x32_10(c, k, out)
        return [int(v) for v in out]

    assert block([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert block([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert block([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_new_entry_points_validate_their_arguments_without_a_gpu():
    """mg_split3_bf16 / mg_dropout / mg_calib_mfma_bf16: the host side refuses bad descriptors before any launch (return code -1 and a
    message), as every other entry point does."""
    import ctypes
    lib = _lib.load()
    desc = (_lib.Split3Desc * 1)()
    desc[0].src, desc[0].rows, desc[0].cols, desc[0].lds = 16, 4, 8, 8
    desc[0].dst, desc[0].ldp, desc[0].order, desc[0].transpose = 32, 8, 7, 0           # order 7 does not exist
    assert lib.mg_split3_bf16(ctypes.cast(desc, ctypes.c_void_p), 1, None) == -1 and 'order 7' in _lib.last_error()
    desc[0].order, desc[0].ldp = 0, 12                                                  # planes must be multiples of 8 columns
    assert lib.mg_split3_bf16(ctypes.cast(desc, ctypes.c_void_p), 1, None) == -1 and 'multiple of 8' in _lib.last_error()
    desc[0].ldp, desc[0].plane_rows = 8, 2                                              # plane_rows goes with order 2 and must cover the rows
    assert lib.mg_split3_bf16(ctypes.cast(desc, ctypes.c_void_p), 1, None) == -1 and 'plane_rows' in _lib.last_error()
    assert lib.mg_split3_bf16(ctypes.cast(desc, ctypes.c_void_p), 0, None) == -1
    assert lib.mg_dropout(16, 16, 8, 0, 1.0, 1, 0, None, None) == -1 and 'p must be in [0, 1)' in _lib.last_error()
    assert lib.mg_dropout(16, 16, 0, 0, 0.5, 1, 0, None, None) == 0                      # nothing to do: no launch
    assert lib.mg_calib_mfma_bf16(None, None, 256, 10, None, None) == -1
    # mg_phone_concat_layer_bf16: at most 16 frame features, their columns inside W, P as wide as N rounded up to 8
    assert lib.mg_phone_concat_layer_bf16(16, 512, 16, 8, 16, 17, 16, 640, 600, None, 512, 1, 32, 512, 0, None) == -1 and 'C=17' in _lib.last_error()
    assert lib.mg_phone_concat_layer_bf16(16, 512, 16, 8, 16, 9, 16, 608, 600, None, 512, 1, 32, 512, 0, None) == -1
    assert lib.mg_segment_sum_feat_bf16(16, 512, 16, 64, 16, 16, 8, 4, 512, 16, 512, 16, 9, 16, 1024, None) == -1 and 'slabs too small' in _lib.last_error()
    assert lib.mg_segment_sum_feat_workspace_bytes(9, 512) % (9 * 512 * 4) == 0
    assert lib.mg_feat_wgrad_reduce(16, 9, 512, 512, 16, 608, 600, 0, None) == -1
    assert lib.mg_f0_tail_rows_f32(16, 130, 16, 16, 16, 16, 16, 16, None, 0, 0, 64, 16, 16, 128, 16, 16, 1 << 20, None) == -1 and 'ldz=130' in _lib.last_error()
    assert lib.mg_f0_tail_rows_f32(16, 128, 16, 16, 16, 16, 16, 16, None, 0, 0, 64, 16, 16, 128, 16, 16, 64, None) == -3     # MG_EWORKSPACE
    assert lib.mg_f0_tail_rows_f32(16, 128, 16, 16, 16, 16, 16, None, None, 3, 20, 64, 16, 16, 128, 16, 16, 1 << 20, None) == -1 and "B x T frames" in _lib.last_error()
    assert lib.mg_f0_tail_rows_f32_workspace_bytes(22528) == 256 * 4164 * 4
    assert lib.mg_phone_mse_rows_f32(16, 0, 16, 16, 8, 16, 16, None) == -1
    assert lib.mg_phone_concat_layer_bf16(16, 96, 16, 8, 16, 9, 16, 609, 600, None, 100, 1, 32, 128, 0, None) == -1 and 'ldp=96' in _lib.last_error()
    for precision in ('fp32', 'bf16', 'bf16x3'):
        from morgana_amd import functional as F_hip
        F_hip.set_precision(precision)
    F_hip.set_precision('fp32')
    with pytest.raises(ValueError):
        F_hip.set_precision('fp16')
    assert F_hip.recurrent_precision('bf16x3') == 'fp32' and F_hip.recurrent_precision('bf16') == 'bf16'


def test_no_cpu_fallback():
    from morgana_amd import utils, losses
    x = torch.zeros(2, 3, 4)
    dur = torch.ones(2, 3, 1, dtype=torch.int64)
    with pytest.raises(_lib.MorganaHipError):
        utils.upsample_to_repetitions(x, dur, max_len=3)
    with pytest.raises(_lib.MorganaHipError):
        losses.mse(torch.zeros(2, 3, 1), torch.zeros(2, 3, 1), torch.tensor([3, 2]))
    with pytest.raises(TypeError):
        utils.upsample_to_repetitions(x, torch.ones(2, 3, 1))          # float durations, as the reference
    from morgana_amd import metrics
    with pytest.raises(_lib.MorganaHipError):                          # streaming metrics: device accumulators only
        metrics.RMSE().accumulate(torch.zeros(2, 3, 1), torch.zeros(2, 3, 1), torch.tensor([3, 2]))
    with pytest.raises(_lib.MorganaHipError):                          # masked Mean (the shipped model's V/UV accuracy) likewise
        metrics.Mean().accumulate(torch.zeros(2, 3, 1), torch.tensor([3, 2]))
    from morgana_amd.viz import synthesis
    with pytest.raises(_lib.MorganaHipError):                          # MLPG: no host solver behind the reference's signature
        synthesis.MLPG(torch.zeros(2, 5, 3), torch.ones(3), padding_size=2, seq_len=torch.tensor([5, 4]))


def test_metrics_handler_collections():
    """morgana/metrics.py:50-186: constructor metrics land in 'all', 'train' and 'valid'; add_metrics('all') reaches every
    collection; one metric object is shared by the collections it was added to; kwargs dicts are passed through."""
    from morgana_amd import metrics
    handler = metrics.Handler(loss=metrics.Mean())
    extra = metrics.Mean()
    handler.add_metrics('all', extra=extra)
    handler.add_metrics('test', only_test=metrics.Mean(hidden=True))
    assert set(handler['train']) == {'loss', 'extra'} and set(handler['test']) == {'extra', 'only_test'}
    assert handler['train']['extra'] is handler['valid']['extra'] is extra
    handler.accumulate('train', loss=torch.tensor([1., 3.]), extra=(torch.tensor([2.]), {'seq_len': None}))
    assert handler.results_as_json_dict('valid') == {'loss': pytest.approx(2.0), 'extra': pytest.approx(2.0)}
    assert 'only_test' not in handler.results_as_json_dict('test')
    handler.reset_state('train')
    assert handler.results_as_json_dict('train', prefix='x_') == {'x_loss': 0.0, 'x_extra': 0.0}
    with pytest.raises(ValueError, match='No collection'):
        handler.accumulate('', loss=torch.tensor([1.]))
    handler.add_collection('synth', from_collections='test')
    assert set(handler['synth']) == {'extra', 'only_test'}
    # the mappings are LIVE, as the reference's dicts (morgana/metrics.py:65-83): a metric written into one is registered
    direct = metrics.Mean()
    handler['train']['direct'] = direct
    handler.metrics.update(via_all=metrics.Mean())
    handler.accumulate('train', direct=torch.tensor([4.]))
    assert handler.result('train')['direct'] == pytest.approx(4.0) and 'via_all' in handler.collections['all']
    assert handler.collections['train'] is handler['train'] and handler.metrics is handler['all']
    # the same name may hold different objects in different collections (:95-101)
    a, b = metrics.Mean(), metrics.Mean()
    handler.add_metrics('train', shared_name=a)
    handler.add_metrics('valid', shared_name=b)
    assert handler['train']['shared_name'] is a and handler['valid']['shared_name'] is b and handler['all']['shared_name'] is b
    with pytest.raises(KeyError):
        handler.accumulate('test', loss=torch.tensor([1.]))           # 'test' holds no metric of that name


def test_order_of_operations_is_a_per_call_choice():
    """utils.upsample_to_repetitions(phone_rate=) records the caller's order of operations on the lazy sequence; None = the process
    default (MORGANA_PHONE_RATE); models carry it as an attribute (base_models.BaseModel.phone_rate) - nothing writes a global."""
    from morgana_amd import models, ops, utils
    assert ops.phone_rate_choice(None) == ops.PHONE_RATE and ops.phone_rate_choice(False) is False and ops.phone_rate_choice(1) is True
    lab, dur = torch.zeros(2, 3, 8), torch.ones(2, 3, dtype=torch.int64)
    seq = utils.upsample_to_repetitions(lab, dur, max_len=3, fused=True, phone_rate=True)     # lazy: no kernel behind this call
    assert isinstance(seq, utils.UpsampledSequence) and seq.phone_rate is True and seq.pending()
    assert models.F0Model().phone_rate is None and models.F0Model(phone_rate=False).phone_rate is False
    assert models.RNNSPSS(phone_rate=True).phone_rate is True
    assert not ops.phone_rate_table_ok(20480, 256000, 512, 128, ops.ACT_SIGMOID, enabled=False)
    assert ops.phone_rate_table_ok(20480, 256000, 512, 128, ops.ACT_SIGMOID, enabled=True)
    assert not ops.phone_rate_gru_ok(4096, 64000, 512, enabled=False) and ops.phone_rate_gru_ok(4096, 64000, 512, enabled=True)


def test_device_batches_shapes_without_a_gpu():
    """data.DeviceBatches: batch count, order and the kept last partial batch (the collate itself needs the device)."""
    import numpy as np
    from morgana_amd import data
    utts = [{'name': 'u%d' % i} for i in range(10)]
    loader = data.DeviceBatches(utts, 4, {}, 'cpu')
    assert len(loader) == 3
    names = [b['name'] for b in loader]                                # only non-numeric features: no kernel involved
    assert names == [['u0', 'u1', 'u2', 'u3'], ['u4', 'u5', 'u6', 'u7'], ['u8', 'u9']]
    shuffled = [n for b in data.DeviceBatches(utts, 4, {}, 'cpu', shuffle=np.random.RandomState(1)) for n in b['name']]
    assert sorted(shuffled) == sorted(u['name'] for u in utts) and shuffled != [u['name'] for u in utts]
    with pytest.raises(ValueError):
        data.DeviceBatches(utts, 0, {}, 'cpu')


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, 'morgana_amd')
    for root, _, files in os.walk(pkg):
        for name in files:
            if name.endswith('.py'):
                text = open(os.path.join(root, name)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), name


def test_state_dict_keys_match_reference_layout():
    from morgana_amd import models
    assert sorted(models.F0Model().state_dict()) == sorted(
        ['layers.%d.%s' % (i, k) for i in (0, 2, 4, 6) for k in ('weight', 'bias')])
    keys = sorted(models.RNNSPSS().state_dict())
    assert 'layers.2.layer.weight_ih_l0' in keys and 'layers.5.bias' in keys


def test_product_library_holds_no_experiments():
    """Measured-slower experiments and timing probes are compiled into the lab builds only: the product
    library's symbol table names none of their kernels, the header documents no value whose results are garbage, and mg_set_tuning
    refuses every key / value that is not a choice between forms with the same results (needs no GPU: host code only)."""
    import re
    import subprocess
    from morgana_amd import _lib
    names = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH], universal_newlines=True)
    for kernel in ('wgrad_fused_solo_kernel', 'wgrad_fused64w_kernel', 'f0_l2tail_split_kernel', 'f0_l2tail_wide_kernel'):
        assert kernel not in names, kernel
    assert not re.search(r'wgrad_fused64_kernelILi3ELi[1-9]', names)              # the PROBE != 0 instantiations
    assert not re.search(r'f0_l2tail_kernelILi[1-9]', names)
    assert not re.search(r'gemm_nt_persist_kernelILi256ELi\dELb1', names)         # the staggered wave groups
    header = open(os.path.join(REPO, 'include', 'morgana_hip.h')).read()
    assert 'results garbage' not in header                                    # no documented value with invalid results
    lib = _lib.load()
    for key, value in ((0, 4), (0, 8), (0, 15), (0, 100), (0, 32), (1, 1), (7, 64), (7, 1), (7, 70), (2, 8), (9, 0), (-1, 0)):
        assert lib.mg_set_tuning(key, value) != 0, (key, value)
    for key, value in ((0, 13), (0, 12), (0, 0), (2, 1), (2, 0), (3, 1), (3, 0), (4, 48), (4, 0), (5, 2), (5, 0), (6, 1), (6, 0), (7, 65), (7, 91), (2, 4), (7, 0), (2, 0)):
        assert lib.mg_set_tuning(key, value) == 0, (key, value)


def test_host_pack_equals_concatenate():
    """mg_host_pack (the loader's host half: utterance arrays back to back into the pinned staging buffer, by host threads) against
    np.concatenate for ragged pieces, every thread count, a zero-length piece; a destination that is too small is refused."""
    import ctypes
    lib = _lib.load()
    rng = np.random.RandomState(3)
    items = [rng.rand(int(n), 37).astype(np.float32) for n in rng.randint(0, 3000, size=41)]
    items[5] = np.zeros((0, 37), np.float32)
    want = np.concatenate(items, axis=0)
    srcs = (ctypes.c_void_p * len(items))(*[a.ctypes.data for a in items])
    sizes = (ctypes.c_int64 * len(items))(*[a.nbytes for a in items])

    def pack(out, capacity, threads):
        return lib.mg_host_pack(ctypes.cast(srcs, ctypes.c_void_p), ctypes.cast(sizes, ctypes.c_void_p), len(items),
                                ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(capacity), threads)

    for threads in (1, 2, 3, 8, 64):
        out = np.full(want.shape[0] * 37 + 16, -1.0, np.float32)
        assert pack(out, out.nbytes, threads) == 0, lib.mg_last_error()
        assert np.array_equal(out[:want.size].reshape(want.shape), want) and np.all(out[want.size:] == -1.0)
    out = np.empty(want.size, np.float32)
    assert pack(out, out.nbytes - 4, 4) != 0 and b'do not fit' in lib.mg_last_error()
    assert pack(out, out.nbytes, 0) != 0 and pack(out, out.nbytes, 65) != 0
