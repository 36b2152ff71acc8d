"""Throughput of the CLI trainer itself (train_uniter.py: dataset files -> collate -> prefetcher -> fused step -> epoch log) on a
synthetic dataset of BASELINE configs[1] shapes, next to bench.py's figure for the bare step: what the input pipeline and the
trainer loop cost.  Usage: python tests/tools/cli_throughput.py [fp32|bf16] [extra train_uniter.py flags]"""
import os, sys, tempfile
sys.path.insert(0, '.')
import train_uniter
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
tmp = tempfile.mkdtemp()
train_uniter.main(['--config', 'uniter-base', '--data_path', tmp + '/data', '--model_path', tmp + '/ckpt', '--vis_path', tmp + '/vis',
                   '--synthetic', os.environ.get('CLI_SAMPLES', '960'), '--batch_size', '16', '--max_epoch', '3', '--lr', '3e-5', '--warmup_steps', '10',
                   '--gradient_accumulation', '1', '--pos_wt', '1.8', '--max_txt_len', '128', '--num_bb', '36', '--seed', '1',
                   '--log_every', '1000', '--hash_tokenizer', '--synthetic_full_length', '--no_model_checkpoints', '--precision', prec] + sys.argv[2:])
