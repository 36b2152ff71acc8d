"""Markdown table of a profile set's bench lines: python tests/tools/results_table.py profiles r04_"""
import json, os, sys
d, pre = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ''
rows = [('bench.json', 'default: config 2, fp32 arithmetic (`--precision fp32x3`)'),
        ('bench_native_fp32.json', '`--precision fp32` (native fp32 MFMA kernels)'),
        ('bench_attention_on_fp32_mfma.json', 'default with `UNITER_ATTN_X3=0` (attention on the fp32 MFMAs)'),
        ('bench_bf16.json', '`--precision bf16` (config 3 arithmetic, 1 GPU)'),
        ('bench_ragged_packed.json', '`--ragged --packed`'),
        ('bench_bf16_ragged_packed.json', '`--precision bf16 --ragged --packed`'),
        ('bench_multitask.json', '`--workload multitask --batch 32` (config 5)'),
        ('bench_bf16_multitask.json', '`--precision bf16 --workload multitask --batch 32`'),
        ('bench_large.json', '`--model large --batch 8 --num_bb 50` (config 4 shapes)'),
        ('bench_large_bf16.json', '`--model large … --precision bf16` (config 4)'),
        ('bench_under_rocprof.json', 'default under `rocprofv3 --kernel-trace --stats`'),
        ('bench_rccl_one_rank_forced.json', 'default, RCCL exchange forced on one rank'),
        ('bench_rccl_one_rank_forced_sparse.json', '… with `--dp_sparse_embeddings`')]
print('| run | samples/s | ms/step | step MFMA | native fp32 kernels, same process |')
print('|---|---|---|---|---|')
for f, name in rows:
    p = os.path.join(d, pre + f)
    if not os.path.exists(p):
        continue
    try:
        j = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print('| %s | ERR %s |' % (name, e)); continue
    nat = j.get('native_fp32') or {}
    print('| %s | %.0f | %.2f | %s | %s |' % (name, j['value'], j['ms_per_step'], j.get('step_mfma_frac', ''),
                                           ('%.0f (%.2f ms): x %.2f' % (nat['value'], nat['ms_per_step'], j['value'] / nat['value'])) if nat else ''))
