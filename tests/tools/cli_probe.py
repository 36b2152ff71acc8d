import json, sys, tempfile, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import train_uniter
from common import TINY
for extra in ([], ['--ragged_regions']):
    for seed in (1, 2, 3):
        tmp = tempfile.mkdtemp()
        cfg = os.path.join(tmp, 'tiny.json'); open(cfg, 'w').write(json.dumps(dict(TINY, vocab_size=28996, max_position_embeddings=64)))
        best, tm = train_uniter.main(['--config', cfg, '--data_path', tmp + '/data', '--model_path', tmp + '/ckpt', '--vis_path', tmp + '/vis',
            '--synthetic', '48', '--batch_size', '8', '--max_epoch', '3', '--lr', '1e-3', '--warmup_steps', '2', '--gradient_accumulation', '2',
            '--pos_wt', '1.8', '--max_txt_len', '16', '--seed', str(seed), '--log_every', '3'] + extra)
        print('PROBE', extra, seed, round(best['aucroc'], 3), flush=True)
