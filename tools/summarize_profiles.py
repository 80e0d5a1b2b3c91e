#!/usr/bin/env python
"""Condense the outputs of tools/gpu_run.sh (gpurun_out/<tag>/) into small tracked files under profiles/.

    python tools/summarize_profiles.py r02 [subdir of gpurun_out]

Writes profiles/<round>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats summary, verbatim)
       profiles/<round>_pmc_summary.json   (per-kernel means of every PMC pass + derived figures)
       profiles/<round>_layer_profile.json (per-layer HIP-event table dumped by bench.py)
       profiles/<round>_bench.json         (the bench line)
PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
(16 B/lane) coalesced reads, so the read side is doubled: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out', sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r01'
os.makedirs(P, exist_ok=True)

for sub in ('rp_drv', 'rp_stats'):             # (tools/gpu_run.sh: rp_drv; the scripts of rounds 1-5: rp_stats)
    if os.path.exists(os.path.join(G, sub, 'drv_kernel_stats.csv')):
        shutil.copy(os.path.join(G, sub, 'drv_kernel_stats.csv'), os.path.join(P, rnd + '_kernel_stats.csv'))
        break
# single files that are kept as they are
for src, dst in (('parity_census_modes.json', '_parity_census.json'), ('parity_census_single_image.json', '_parity_census_single_image.json'),
                 ('rect_batches.json', '_rect_batches.json'), ('mixed_batch.json', '_mixed_batch.json'),
                 ('precise_priority_ab.json', '_precise_priority_ab.json'), ('block_timing.json', '_block_timing.json')):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, rnd + dst))
if os.path.exists(os.path.join(G, 'parity_census_368x496.json')) and os.path.exists(os.path.join(G, 'parity_census_496x368.json')):
    json.dump({'368x496': json.load(open(os.path.join(G, 'parity_census_368x496.json')))['paths'],
               '496x368': json.load(open(os.path.join(G, 'parity_census_496x368.json')))['paths']},
              open(os.path.join(P, rnd + '_parity_census_landscape.json'), 'w'), indent=1)
if os.path.exists(os.path.join(G, 'rp_bench', 'bench_kernel_stats.csv')):      # rocprofv3 --kernel-trace --stats -- python bench.py
    shutil.copy(os.path.join(G, 'rp_bench', 'bench_kernel_stats.csv'), os.path.join(P, rnd + '_bench_kernel_stats.csv'))

for sub, stem, dst in (('rp_b1', 'b1', '_b1_kernel_stats.csv'), ('rp_rect', 'rect', '_rect_368x496_kernel_stats.csv'), ('rp_precise', 'precise', '_precise_kernel_stats.csv'),
                       ('rp_mixed', 'mixed', '_mixed_batch_kernel_stats.csv')):
    f = os.path.join(G, sub, stem + '_kernel_stats.csv')
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, rnd + dst))
if os.path.exists(os.path.join(G, 'parity_census.json')) and not os.path.exists(os.path.join(G, 'parity_census_modes.json')):
    shutil.copy(os.path.join(G, 'parity_census.json'), os.path.join(P, rnd + '_parity_census.json'))
fg = os.path.join(G, 'bench_force_gather.log')
if os.path.exists(fg):
    lines = [l for l in open(fg) if l.startswith('{')]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(P, rnd + '_bench_force_gather.json'), 'w'), indent=1)

def pmc_summary(prefix, skip, batch, out_suffix):
  summary = {'source': 'rocprofv3 --pmc <counters> --kernel-trace -- python tools/profile_driver.py --batch %d ' % batch +
                       '(one pass per counter group); rocprofv3 --kernel-trace --stats for durations',
             'units': {'FETCH_SIZE': 'KiB (uncorrected)', 'WRITE_SIZE': 'KiB', 'SQ_*': 'summed over the chip',
                       'GRBM_GUI_ACTIVE': 'summed over 8 XCDs'},
             'kernels': {}}
  for d in sorted(os.listdir(G)):
      f = os.path.join(G, d, 'drv_counter_collection.csv')
      if not d.startswith(prefix) or (skip and d.startswith(skip)) or not os.path.exists(f):
          continue
      agg = collections.defaultdict(lambda: collections.defaultdict(list))
      for r in csv.DictReader(open(f)):
          agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
          agg[r['Kernel_Name']]['_dur_ns_' + d].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
      for k, cs in agg.items():
          if not any(t in k for t in ('conv_mfma', 'conv_wino', 'conv1_wino', 'conv1_fused', 'conv_bf16x3', 'conv1x1_pair', 'conv3x3_c3', 'conv_splitk', 'prep', 'resize_cubic')) and not k.startswith('pp_'):
              continue
          e = summary['kernels'].setdefault(k, {})
          for c, vals in cs.items():
              e[c] = {'n': len(vals), 'mean': statistics.mean(vals), 'median': statistics.median(vals), 'min': min(vals), 'max': max(vals)}

  N_SIMD = 256 * 4
  for k, e in summary['kernels'].items():
      der = {}
      if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
          der['hbm_bytes_per_launch_min'] = (2 * e['FETCH_SIZE']['min'] + e['WRITE_SIZE']['min']) * 1024
          der['hbm_bytes_per_launch_mean'] = (2 * e['FETCH_SIZE']['mean'] + e['WRITE_SIZE']['mean']) * 1024
          der['hbm_bytes_per_launch_median'] = (2 * e['FETCH_SIZE']['median'] + e['WRITE_SIZE']['median']) * 1024
      if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and 'GRBM_GUI_ACTIVE' in e:
          # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs
          der['mfma_busy_frac'] = (e['SQ_VALU_MFMA_BUSY_CYCLES']['mean'] / N_SIMD) / (e['GRBM_GUI_ACTIVE']['mean'] / 8)
          dur = e.get('_dur_ns_' + prefix + 'SQ_VALU_MFMA_BUSY_CYCLES')
          if dur:
              der['effective_clock_ghz'] = e['GRBM_GUI_ACTIVE']['mean'] / 8 / dur['mean']
      if 'SQ_LDS_BANK_CONFLICT' in e and 'SQ_LDS_IDX_ACTIVE' in e and e['SQ_LDS_IDX_ACTIVE']['mean'] > 0:
          der['lds_conflict_frac'] = e['SQ_LDS_BANK_CONFLICT']['mean'] / e['SQ_LDS_IDX_ACTIVE']['mean']
      if 'SQ_WAVE_CYCLES' in e:
          w = e['SQ_WAVE_CYCLES']['mean']
          for c in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY'):
              if c in e:
                  der[c.lower() + '_frac'] = e[c]['mean'] / w
      e['derived'] = der
  if summary['kernels']:
      json.dump(summary, open(os.path.join(P, rnd + out_suffix), 'w'), indent=1)


pmc_summary('pmc_', 'pmc_b1_', 32, '_pmc_summary.json')
pmc_summary('pmc_b1_', None, 1, '_pmc_b1_summary.json')

for src, dst in (('prof_bench.json', '_layer_profile.json'),):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, rnd + dst))
bl = os.path.join(G, 'bench.log')
if os.path.exists(bl):
    lines = [l for l in open(bl) if l.startswith('{')]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(P, rnd + '_bench.json'), 'w'), indent=1)
if os.path.exists(os.path.join(P, rnd + '_pmc_summary.json')):
    for k, e in json.load(open(os.path.join(P, rnd + '_pmc_summary.json')))['kernels'].items():
        print(k[:60], json.dumps(e['derived']))

# the bench line's roofline, recomputed from the rocprofv3 kernel stats of the same command: issued / algorithmic FLOP per launch (from the
# bench line = the engine's launch plan) over rocprofv3's average duration of that kernel
bj, ks = os.path.join(P, rnd + '_bench.json'), os.path.join(P, rnd + '_bench_kernel_stats.csv')
if os.path.exists(bj) and os.path.exists(ks):
    b = json.load(open(bj))
    roof = b.get('roofline') or {}
    rows = [r for r in csv.DictReader(open(ks)) if roof.get('kernel') and roof['kernel'] in r['Name']]
    if rows:
        calls = sum(int(r['Calls']) for r in rows)
        avg_ns = sum(float(r['TotalDurationNs']) for r in rows) / calls
        peak = roof['peak']
        chk = {'kernel': roof['kernel'], 'rocprofv3_calls': calls, 'rocprofv3_avg_launch_ms': avg_ns / 1e6, 'bench_avg_launch_ms': roof['avg_launch_ms'],
               'frac_from_rocprofv3': roof['issued_flop_per_launch'] / (avg_ns * 1e-9) / 1e12 / peak, 'frac_in_bench_line': roof['frac'],
               'algorithmic_frac_from_rocprofv3': roof['flop_per_launch'] / (avg_ns * 1e-9) / 1e12 / peak,
               'algorithmic_frac_in_bench_line': roof.get('algorithmic_frac'),
               'note': 'rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras (warm-up, timed and the one '
                       'fully instrumented step; the calibration forward at batch 1 uses other kernels)'}
        json.dump(chk, open(os.path.join(P, rnd + '_roofline_check.json'), 'w'), indent=1)
        print('roofline check', json.dumps(chk))
