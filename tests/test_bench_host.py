"""CPU: the host-side bookkeeping of bench.py -- profile label -> HIP kernel mapping (what the live HIP-event figures and the committed
rocprofv3 / PMC summaries are joined on), grouping of the dominant kernel, issued-vs-algorithmic roofline pair."""
import importlib.util
import os

from conftest import ROOT

spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_profile_labels_map_to_hip_kernels():
    m = bench.rocprof_kernel
    assert m('conv_wino_f2x2_7x7r/t2') == m('conv_wino_f2x2_7x7r/t3') == m('conv_wino_f2x2_7x7r') == 'conv_wino_kernel<7, 0, 0, 1>'
    assert m('conv_wino_f2x2_7x7r/t2:units') == 'conv_wino_kernel<7, 0, 1, 1>'
    # merged tails (label ".../t<g>m"): the full runs are the same kernel, the units the merged-tail instantiation
    assert m('conv_wino_f2x2_7x7r/t1m') == m('conv_wino_f2x2_7x7r/t2m') == 'conv_wino_kernel<7, 0, 0, 1>'
    assert m('conv_wino_f2x2_7x7r/t1m:units') == 'conv_wino_kernel<7, 0, 1, 3>' and m('conv_wino_f2x2_3x3r/t6m:combine') == 'conv_wino_tail_reduce_kernel'
    assert m('conv_wino_f2x2_3x3r/t8:combine') == 'conv_wino_tail_reduce_kernel'
    assert m('conv_wino_f2x2_7x7/u1') == 'conv_wino_kernel<7, 0, 1, 0>'
    assert m('conv_wino_f2x2_3x3') == 'conv_wino_kernel<3, 0, 0, 0>'
    assert m('conv7x7_v6_t17x32_n128') == 'conv_mfma_v6_kernel<7, 17, 0>'
    assert m('conv3x3_v5_t8x16_n64').startswith('conv_mfma_v5_kernel<3, 8, 16, 64,')
    assert m('pp_peaks') == 'pp_peaks'
    # one half of a batch cut in two by images: the same kernels
    assert m('conv_wino_f2x2_7x7@0+5') == 'conv_wino_kernel<7, 0, 0, 0>' and m('conv_wino_f2x2_7x7/u2@5+3') == 'conv_wino_kernel<7, 0, 1, 0>'
    assert m('conv_wino_f2x2_3x3r/t3m:units@0+22') == 'conv_wino_kernel<3, 0, 1, 3>'


def test_dominant_kernel_groups_labels_of_one_hip_kernel_and_keeps_issued_below_algorithmic():
    prof = [dict(layer='Mconv2_stage2', kernel='conv_wino_f2x2_7x7r/t2', total_ms=8.0, launches=10, flop_per_launch=200e9, issued_flop_per_launch=200e9 * 100 / 196),
            dict(layer='Mconv1_stage2', kernel='conv_wino_f2x2_7x7r/t3', total_ms=6.0, launches=5, flop_per_launch=300e9, issued_flop_per_launch=300e9 * 100 / 196),
            dict(layer='conv1_1+conv1_2', kernel='conv1_fused_t8x16_n64', total_ms=10.0, launches=4, flop_per_launch=330e9, issued_flop_per_launch=330e9)]
    name, ms, n, flop, issued, labels = bench.dominant_kernel(prof)
    assert name == 'conv_wino_kernel<7, 0, 0, 1>' and n == 15 and abs(ms - 14.0) < 1e-9
    assert labels == ['conv_wino_f2x2_7x7r/t2', 'conv_wino_f2x2_7x7r/t3']
    assert abs(issued / flop - 100 / 196) < 1e-12
    frac = issued / (ms * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS
    assert frac < 1.0 < flop / (ms * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS


def test_pick_device_per_rank():
    """RCCL: LOCAL_RANK when the rank sees the node's GPUs, device 0 under a launcher that masks one GPU per rank, a loud exit otherwise;
    gloo smoke mode: ranks wrap around the visible GPUs."""
    import pytest
    assert [bench.pick_device(r, 8, 8, 'nccl') for r in range(8)] == list(range(8))
    masked = {'HIP_VISIBLE_DEVICES': '3'}
    assert [bench.pick_device(r, 1, 8, 'nccl', env=masked) for r in range(8)] == [0] * 8          # HIP_VISIBLE_DEVICES = one device per rank
    with pytest.raises(SystemExit):
        bench.pick_device(3, 1, 8, 'nccl', env={})                                      # one unmasked GPU for 8 RCCL ranks: all on one device
    assert bench.pick_device(0, 1, 1, 'nccl') == 0
    with pytest.raises(SystemExit):
        bench.pick_device(5, 4, 8, 'nccl')                                              # 4 GPUs for 8 ranks: no
    with pytest.raises(SystemExit):
        bench.pick_device(1, 1, 1, 'nccl')
    assert [bench.pick_device(r, 1, 8, 'gloo') for r in range(8)] == [0] * 8
    assert [bench.pick_device(r, 2, 4, 'gloo') for r in range(4)] == [0, 1, 0, 1]


def test_compare_records_counts_poses_exact_scores_to_tolerance():
    """bench.compare_records (the self-validation of an N > 1 line): identical counts / poses + scores within 1e-5 -> match; a pose that
    moved, a lost person, or a score off by more than the tolerance -> no match, with the first offending record named; person
    capacities of the two arrays may differ (a rank whose capacity grew)."""
    import importlib
    import numpy as np
    native = importlib.import_module(bench.PKG + '.native')
    rng = np.random.default_rng(0)

    def make(cap, n_people):
        r = np.zeros(len(n_people), native.result_dtype(cap))
        for i, n in enumerate(n_people):
            r[i]['n_people'] = n
            r[i]['n_peaks'] = 10 * n + i
            r[i]['poses'][:n] = rng.integers(0, 300, (n, 18, 3))
            r[i]['scores'][:n] = rng.random(n) * 30
        return r
    a = make(64, [3, 0, 5, 1])
    b = np.zeros(4, native.result_dtype(128))
    for f in ('n_people', 'n_peaks', 'status', 'n_subsets_raw'):
        b[f] = a[f]
    b['poses'][:, :64] = a['poses']
    b['scores'][:, :64] = a['scores']
    ok = bench.compare_records(a, b)
    assert ok['shard_records_match'] and ok['records_compared'] == 4 and ok['bitwise_equal_records'] == 4 and ok['max_abs_score_diff'] == 0.0
    b['scores'][2, 1] += 3e-6
    near = bench.compare_records(a, b)
    assert near['shard_records_match'] and near['bitwise_equal_records'] == 3 and 2e-6 < near['max_abs_score_diff'] < 4e-6
    b['scores'][2, 1] += 1e-3
    assert not bench.compare_records(a, b)['shard_records_match']
    b['scores'][2, 1] = a['scores'][2, 1]
    b['poses'][0, 2, 5, 0] += 1.0
    bad = bench.compare_records(a, b)
    assert not bad['shard_records_match'] and bad['first_mismatch']['record'] == 0 and bad['records_with_identical_counts_and_poses'] == 3
    b['poses'][0] = 0
    b['poses'][0, :64] = a['poses'][0]
    b['n_people'][3] = 0
    lost = bench.compare_records(a, b)
    assert not lost['shard_records_match'] and lost['first_mismatch'] == {'record': 3, 'n_people': [1, 0], 'n_peaks': [13, 13], 'status': [0, 0]}
    assert not bench.compare_records(a[:0], b[:0])['shard_records_match']            # nothing compared is not a match


def test_hardware_queue_default_follows_the_rank_count(monkeypatch):
    # one rank: the package's default (2 queues: detect_precise's lanes); several ranks / the one-rank RCCL group: 8 (RCCL's streams apart)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    f = bench._multi_rank_argv
    assert not f([]) and not f(['--gpus', '1']) and not f(['--steps', '5', '--gpus=1'])
    assert f(['--gpus', '2']) and f(['--gpus=8', '--steps', '3']) and f(['--force-gather'])
    monkeypatch.setenv('WORLD_SIZE', '4')
    assert f([]) and f(['--gpus', '4'])
