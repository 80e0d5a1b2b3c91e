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
    assert [bench.pick_device(r, 1, 8, 'nccl') for r in range(8)] == [0] * 8          # HIP_VISIBLE_DEVICES = one device per rank
    assert bench.pick_device(0, 1, 1, 'nccl') == 0
    with pytest.raises(SystemExit):
        bench.pick_device(5, 4, 8, 'nccl')                                              # 4 GPUs for 8 ranks: no
    with pytest.raises(SystemExit):
        bench.pick_device(1, 1, 1, 'nccl')
    assert [bench.pick_device(r, 1, 8, 'gloo') for r in range(8)] == [0] * 8
    assert [bench.pick_device(r, 2, 4, 'gloo') for r in range(4)] == [0, 1, 0, 1]
