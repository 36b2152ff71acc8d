#!/usr/bin/env python
"""CLI of the UNITER fine-tuning path on MI355X: counterpart of the reference's
train_uniter.py (same flags, same hooks, same model/kwargs plumbing), running on
libuniter_hip.so.

    python train_uniter.py --config config/uniter-base.json --data_path ./dataset --feature_path ./dataset/img_feats \
        --pretrained_model_file uniter-base.pt --lr 3e-5 --scheduler warmup_cosine --warmup_steps 500 \
        --max_epoch 30 --batch_size 16 --gradient_accumulation 2 --pos_wt 1.8 --seed 43
    torchrun --nproc-per-node 8 train_uniter.py ... --parallel_computing True      # DP over RCCL

Additive flags (not in the reference): --synthetic N (write a synthetic dataset in the reference's
on-disk format under --data_path and train on it), --hash_tokenizer (offline tokenizer).
"""
import argparse
import json
import os
from functools import partial

import torch
from torch.utils import data

from meme_challenge_amd.data import (MemeDataset, ConfounderSampler, HashTokenizer, write_synthetic_dataset,
                                     build_feature_shard, DevicePrefetcher)
from meme_challenge_amd.meme_uniter import MemeUniter
from meme_challenge_amd.model import UniterModel, UniterConfig, resolve_config
from meme_challenge_amd.train_template import TrainerTemplate, LOGGER
from meme_challenge_amd import dp


class RankShard(data.Sampler):
    """One rank's share of the epoch order an inner sampler draws.  Every rank runs the same inner sampler with
    the same seed (set_seed + python's `random` in ConfounderSampler), so the orders agree; rank r takes every
    world-th index, padded by wrap-around so that all ranks step equally often."""

    def __init__(self, inner, rank, world):
        self.inner, self.rank, self.world = inner, rank, world

    def __iter__(self):
        return iter(dp.shard_indices(list(self.inner), self.rank, self.world))

    def __len__(self):
        return (len(self.inner) + self.world - 1) // self.world

IMG_DIM = 2048            # utils/const.py


class TrainerUniter(TrainerTemplate):

    def init_model(self):
        if self.pretrained_model_file:
            checkpoint = torch.load(self.pretrained_model_file, map_location='cpu')
            LOGGER.info('Using pretrained UNITER base model {}'.format(self.pretrained_model_file))
            # the reference goes through UniterForPretraining and keeps `.uniter` (train_uniter.py:26-33)
            sd = {k[len('uniter.'):]: v for k, v in checkpoint['model_state_dict'].items() if k.startswith('uniter.')}
            base = UniterModel.from_pretrained(self.config['config'], state_dict=sd, img_dim=IMG_DIM)
            self.model = MemeUniter(uniter_model=base, hidden_size=base.config.hidden_size,
                                    n_classes=self.config['n_classes'])
        else:
            self.load_model()
        self._apply_runtime_options()

    def _apply_runtime_options(self):
        """--precision / --pack_padded belong to the encoder object: every place that builds one applies them, so the
        final reload in end_training scores with the arithmetic that was trained and validated."""
        self.model.uniter_model.pack_padded = bool(self.config.get('pack_padded', False))
        prec = self.config.get('precision', 'fp32x3')
        cfg = self.model.uniter_model.config
        if prec == 'fp32x3' and (cfg.hidden_size % 32 or cfg.intermediate_size % 32):
            LOGGER.info('precision fp32x3 needs hidden / intermediate sizes %% 32 == 0: this model runs the native fp32 kernels')
            prec = 'fp32'
        self.model.uniter_model.precision = prec

    def load_model(self):
        uniter_config = resolve_config(self.config['config'])
        uniter_model = UniterModel(uniter_config, img_dim=IMG_DIM)
        self.model = MemeUniter(uniter_model=uniter_model, hidden_size=uniter_model.config.hidden_size,
                                n_classes=self.config['n_classes'])
        if self.model_file and os.path.isfile(self.model_file):
            checkpoint = torch.load(self.model_file, map_location='cpu')
            LOGGER.info('Using UNITER model {}'.format(self.model_file))
            self.model.load_state_dict(checkpoint['model_state_dict'])
        self._apply_runtime_options()

    def _forward(self, batch):
        return self.model(img_feat=batch['img_feat'], img_pos_feat=batch['img_pos_feat'],
                          input_ids=batch['input_ids'], position_ids=batch['position_ids'],
                          attention_mask=batch['attn_mask'], gather_index=batch['gather_index'],
                          output_all_encoded_layers=False, seq_lens=batch.get('seq_lens'))

    def eval_iter_step(self, iters, batch, test):
        self.calculate_loss(self._forward(batch), batch['labels'], grad_step=False)

    def train_iter_step(self):
        self.preds = self._forward(self.batch)
        self.calculate_loss(self.preds, self.batch['labels'], grad_step=True)

    def test_iter_step(self, batch):
        return self._forward(batch).squeeze()


def build_parser():
    parser = argparse.ArgumentParser()
    TrainerTemplate.add_default_argparse(parser)
    parser.add_argument('--config', type=str, default='./config/uniter-base.json')
    parser.add_argument('--feature_path', type=str, default='./dataset/img_feats')
    parser.add_argument('--max_txt_len', type=int, default=60)
    parser.add_argument('--conf_th', type=float, default=0.2)
    parser.add_argument('--max_bb', type=int, default=100)
    parser.add_argument('--min_bb', type=int, default=10)
    parser.add_argument('--num_bb', type=int, default=36)
    parser.add_argument('--fc_dim', type=int, default=64)
    parser.add_argument('--dropout', type=float, default=0.2)
    # additive
    parser.add_argument('--synthetic', type=int, default=0, help='write + use a synthetic dataset of N samples per split')
    parser.add_argument('--synthetic_full_length', action='store_true',
                        help='synthetic captions that fill --max_txt_len and images with --num_bb regions (BASELINE configs[1] shapes)')
    parser.add_argument('--hash_tokenizer', action='store_true', help='offline tokenizer instead of bert-base-cased')
    parser.add_argument('--feature_shards', action='store_true',
                        help='pack the per-sample region-feature files of every split into one memory-mapped shard '
                             '(built next to the jsonl on first use) and read from it')
    parser.add_argument('--no_prefetch', action='store_true', help='copy each batch to the GPU synchronously (default: one batch ahead on a side stream)')
    parser.add_argument('--pack_padded', action='store_true', help='token packing: compute the valid positions only')
    parser.add_argument('--ragged_regions', action='store_true',
                        help="mask every sample at its own region count (the reference's collate counts the zero-padded rows of the batch: data.MemeDataset)")
    parser.add_argument('--precision', type=str, default='fp32x3', choices=['fp32', 'fp32x3', 'bf16', 'bf16_hybrid'],
                        help="GEMM arithmetic: fp32x3 (default, the path bench.py times: the reference's fp32 results from six bf16 "
                             "MFMA products per block on three bf16 pieces per value -- no less accurate than the fp32 MFMA kernels, "
                             "~30 %% faster steps), fp32 (native fp32 MFMA kernels), bf16 (bf16-resident operands) or bf16_hybrid")
    return parser


def _cap_cpu_threads():
    """The host side of this trainer is the collate's handful of small tensor operations; with torch's default intra-op
    pool (one thread per core: 128 on a 256-core host) each of them pays the pool's wake-up and a batch takes 38 ms to
    assemble instead of 2.3 ms with eight threads (MI355X box, 16 samples, 36 regions) -- seven times the bf16 training
    step.  UNITER_CPU_THREADS overrides."""
    want = int(os.environ.get('UNITER_CPU_THREADS', '8'))
    had = torch.get_num_threads()
    if want > 0 and had > want:
        torch.set_num_threads(want)
    return had


def main(argv=None):
    had_threads = _cap_cpu_threads()
    try:
        return _main(argv)
    finally:
        torch.set_num_threads(had_threads)       # a caller that runs main() in its own process keeps its pool


def _main(argv=None):
    args, _ = build_parser().parse_known_args(argv)
    config = args.__dict__
    if config['parallel_computing'] and 'RANK' in os.environ and not torch.distributed.is_initialized():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        # RCCL; UNITER_DIST_BACKEND=gloo lets several ranks share one GPU (tests: RCCL wants one device per rank)
        from meme_challenge_amd import dp as _dp
        _dp.prepare_rccl_env(int(os.environ.get('WORLD_SIZE', '1')))       # (UNITER_DP_CAP_CHANNELS=1: RCCL's channels <= the CUs the matrix kernels leave free)
        torch.distributed.init_process_group(os.environ.get('UNITER_DIST_BACKEND', 'nccl'))
    ddp = torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    rank = torch.distributed.get_rank() if ddp else 0
    world = torch.distributed.get_world_size() if ddp else 1
    if config['synthetic'] > 0:
        if rank == 0:                    # one writer; the other ranks wait below before they open any file
            os.makedirs(config['data_path'], exist_ok=True)
            if not os.path.isfile(os.path.join(config['data_path'], 'train.jsonl')):
                full = config.get('synthetic_full_length')
                write_synthetic_dataset(config['data_path'], n=config['synthetic'],
                                        splits=('train', 'dev_seen', 'test_seen'),
                                        text_words=(config['max_txt_len'], config['max_txt_len']) if full else (3, 8),
                                        num_bb=(config['num_bb'], config['num_bb']) if full else (10, 36))
        if ddp:
            torch.distributed.barrier()
        config['feature_path'] = os.path.join(config['data_path'], 'img_feats')
    config = TrainerTemplate.preprocess_args(config)
    if config['hash_tokenizer'] or config['synthetic'] > 0:
        tokenizer = HashTokenizer(max_length=config['max_txt_len'])
    else:
        from transformers import BertTokenizer
        tokenizer = BertTokenizer.from_pretrained('bert-base-cased')      # needs the hub cache
    tokenizer_func = partial(tokenizer, max_length=config['max_txt_len'], padding='max_length', truncation=True,
                             return_tensors='pt', return_length=True)

    def make(fname, train=False, ids=False):
        path = os.path.join(config['data_path'], fname)
        shard = None
        if config['feature_shards']:
            shard = os.path.splitext(path)[0] + '_shard'
            if rank == 0 and not os.path.isfile(shard + '.index.json'):
                with open(path) as f:
                    build_feature_shard(config['feature_path'], [json.loads(l)['id'] for l in f if l.strip()], shard)
            if ddp:
                torch.distributed.barrier()
        ds = MemeDataset(filepath=path, feature_dir=config['feature_path'], feature_shard=shard,
                         text_padding=tokenizer_func, return_ids=ids, confidence_threshold=config['object_conf_thresh'],
                         ragged_regions=config['ragged_regions'])
        kw = dict(batch_size=config['batch_size'], num_workers=config['num_workers'], collate_fn=ds.get_collate_fn(),
                  pin_memory=True)
        if config['num_workers'] > 0:
            # worker processes live across epochs: re-spawning them costs seconds per epoch and loader (an epoch of this trainer
            # is a fraction of a second of GPU time)
            kw.update(persistent_workers=True, prefetch_factor=4)
        if train:
            sampler = ConfounderSampler(ds, config['confounder_repeat'])
            if ddp:                      # the reference's nn.DataParallel split every batch over the GPUs; here every rank draws its own batches
                sampler = RankShard(sampler, rank, world)
            loader = data.DataLoader(ds, sampler=sampler, **kw)
        else:                            # validation / test: every rank scores everything (identical early-stopping decisions)
            loader = data.DataLoader(ds, **kw)
        return loader if config['no_prefetch'] else DevicePrefetcher(loader, config['device'])

    config['test_loader'] = [make(f, ids=True) for f in ('test_seen.jsonl', 'test_unseen.jsonl', 'dev_seen.jsonl',
                                                        'dev_unseen.jsonl')
                             if os.path.isfile(os.path.join(config['data_path'], f))]
    if config['num_folds'] == 0:
        config['train_loader'] = make('train.jsonl', train=True)
        config['val_loader'] = make('dev_seen.jsonl')
        return TrainerUniter(config).train_main()
    # --num_folds k (-1 = all): one run per cross-validation fold, then the ensemble of the folds' predictions
    # (train_uniter.py:183-188 -> utils/crossval.py:132-215 in the reference)
    from meme_challenge_amd.crossval import train_crossval
    split = lambda path: os.path.relpath(path, config['data_path'])
    funcs = {'train': lambda path: make(split(path), train=True), 'val': lambda path: make(split(path)),
             'test': lambda path: make(split(path), ids=True)}
    return train_crossval(TrainerUniter, config, funcs, num_folds=config['num_folds'],
                          dev_size=config['crossval_dev_size'], use_dev_set=config['crossval_use_dev'])


if __name__ == '__main__':
    main()
