"""Training driver around the HIP hot path.

What is kept from the reference's `TrainerTemplate` (train_template.py:26-552) is its CONTRACT, because
`train_uniter.py`-style subclasses and the cross-validation driver are written against it:
  * the hooks `init_model / load_model / train_iter_step / eval_iter_step / test_iter_step`,
    the attributes they use (`self.model`, `self.batch`, `self.iters`, `self.preds`, `self.config`,
    `self.model_file`, `self.pretrained_model_file`) and `calculate_loss(preds, labels, grad_step)`;
  * the public entry points `train_main`, `eval_model`, `export_val_predictions`,
    `export_test_predictions`, `export_metrics`;
  * the CLI flags (`add_default_argparse`, same names / types / defaults) and `preprocess_args`;
  * the artefacts: `{'model_state_dict': ...}` checkpoints (utils/save.py:57-63), `id,proba,label[,gt]`
    prediction CSVs, `<name>_metrics.json`, early stopping on `optimize_for`.

The loop itself is this package's own:
  * per-iteration probabilities, labels and losses are written into preallocated DEVICE buffers
    (`EpochLog`) and copied to the host once per epoch -- no `.item()` / `.cpu()` per iteration;
  * averaging, clipping, the optimizer update and zero_grad are one fused launch behind the
    data-parallel exchange (`trainer.sync_step`);
  * `--parallel_computing` means one process per GPU over RCCL (launch with torchrun): every rank
    trains on its own shard of each epoch's sample order, evaluates the full validation set (so
    early stopping decides identically everywhere), rank 0 writes files, and the ranks leave
    together (barrier) before anything downstream reads those files;
  * `loss_func`: 'bce_logits' (one logit, pos_weight) is one fused kernel; 'bce' and 'ce'
    (train_template.py:66-69) are formed by torch on the model's output and backpropagate into the
    same HIP backward; `--optimizer adamax | sgd` use torch's update on the flat buffers
    (trainer.TorchOptimizerStep) -- the fused step is built for the recipe the reference trains with.
"""
import datetime
import json
import logging
import os
import time
from collections import defaultdict

import torch
import torch.distributed as dist

from . import dp
from .metrics import standard_metrics, find_optimal_threshold
from .trainer import bce_with_logits_loss, get_optimizer, get_scheduler, sync_step
from .utils import set_seed

LOGGER = logging.getLogger('TrainerLogger')
logging.basicConfig(format='%(asctime)s : %(levelname)s - %(message)s', datefmt='%d/%m/%Y %I:%M:%S %p',
                    level=logging.INFO)

METRIC_TAGS = (('F1', 'F1'), ('Precision', 'precision'), ('Recall', 'recall'), ('Accuracy', 'accuracy'),
               ('AUC-ROC', 'aucroc'))


def _distributed():
    return dist.is_available() and dist.is_initialized()


def _is_main():
    return not _distributed() or dist.get_rank() == 0


class _NullWriter(object):
    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def _make_writer(path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(path)
    except Exception:           # tensorboard not installed
        return _NullWriter()


class ModelSaver(object):
    """Checkpoint format of utils/save.py:53-63: torch.save({'model_state_dict': cpu tensors})."""

    def __init__(self, output_path):
        self.output_path = output_path

    def save(self, model, optimizer=None):
        if optimizer is not None and hasattr(optimizer, 'join'):
            optimizer.join()                 # an update overlapped with the next forward may still be running
        tensors = {k: (v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v)
                   for k, v in model.state_dict().items()}
        torch.save({'model_state_dict': tensors}, self.output_path)


class EpochLog(object):
    """Device-side record of one pass over a loader: probabilities, labels, per-iteration losses (and
    sample ids when the dataset returns them).  Buffers grow geometrically and are reused across
    epochs; `collect()` is the only host synchronisation."""

    def __init__(self, device):
        self.device = device
        self._buf = {}
        self.reset()

    def reset(self):
        self.n = 0              # samples
        self.k = 0              # iterations
        self.k_short = 0        # iterations since the last 'log_every' flush

    def _room(self, name, need, dtype):
        b = self._buf.get(name)
        if b is None or b.numel() < need or b.dtype != dtype:
            nb = torch.empty(max(need, 2 * (b.numel() if b is not None else 512)), dtype=dtype, device=self.device)
            if b is not None and b.dtype == dtype:
                nb[:b.numel()] = b
            self._buf[name] = b = nb
        return b

    def add(self, probs, labels, loss, ids=None):
        m = probs.numel()
        self._room('probs', self.n + m, torch.float32)[self.n:self.n + m] = probs.detach().reshape(-1)
        self._room('labels', self.n + m, torch.long)[self.n:self.n + m] = labels.detach().reshape(-1)
        if ids is not None:
            self._room('ids', self.n + m, torch.long)[self.n:self.n + m] = ids.detach().reshape(-1)
        self._room('loss', self.k + 1, torch.float32)[self.k] = loss.detach()
        self.n += m
        self.k += 1
        self.k_short += 1

    def recent_loss(self):
        """Mean loss of the iterations since the previous call (one scalar read, at the logging interval)."""
        v = self._buf['loss'][self.k - self.k_short:self.k].mean().item() if self.k_short else float('nan')
        self.k_short = 0
        return v

    def collect(self, with_ids=False):
        if self.n == 0:
            empty = torch.zeros(0)
            return (empty, torch.zeros(0, dtype=torch.long), 0.0) + ((None,) if with_ids else ())
        probs = self._buf['probs'][:self.n].cpu()
        labels = self._buf['labels'][:self.n].cpu()
        mean_loss = self._buf['loss'][:self.k].mean().item()
        if not with_ids:
            return probs, labels, mean_loss
        ids = self._buf['ids'][:self.n].cpu() if 'ids' in self._buf else None
        return probs, labels, mean_loss, ids


class Plateau(object):
    """Early stopping on one scalar: `better` = improvement over the best so far; stop after `patience`
    evaluations that did not improve it by at least `min_delta`."""

    def __init__(self, maximise, patience, min_delta, start):
        self.sign = 1.0 if maximise else -1.0
        self.patience, self.min_delta = patience, min_delta
        self.best = start
        self.stale = 0

    def update(self, value):
        gain = self.sign * (value - self.best)
        improved = gain > 0
        if improved:
            self.best = value
        self.stale = 0 if gain >= self.min_delta else self.stale + 1
        return improved, self.stale >= self.patience


class TrainerTemplate(object):

    def __init__(self, config):
        self.config = config
        self.device = config.get('device', torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))))
        if not isinstance(config['test_loader'], list):
            config['test_loader'] = [config['test_loader']]
        self.model_file = os.path.join(config['model_path'], config['model_save_name'])
        self.pretrained_model_file = None
        if config.get('pretrained_model_file') is not None:
            self.pretrained_model_file = os.path.join(config['model_path'], config['pretrained_model_file'])
        self.start_epoch = 1
        self.total_iters = 0
        self.terminate_training = False
        self.best_val_metrics, self.train_metrics, self.test_metrics = defaultdict(int), {}, {}
        self.best_val_loss = 1000
        self.train_loss = float('nan')
        self.grad_sync = None
        self._ids = None
        self.log = EpochLog(self.device)
        key = config['optimize_for']
        self.plateau = Plateau(maximise=(key != 'loss'), patience=config['patience'],
                               min_delta=config['early_stop_thresh'], start=(1000 if key == 'loss' else 0))
        self.init_training_params()

    # ------------------------------------------------------------------ set-up
    def init_training_params(self):
        loss_func = self.config['loss_func']
        if loss_func not in ('bce_logits', 'bce', 'ce'):
            raise ValueError('invalid loss_func %r' % loss_func)
        self.init_model()
        self.model.to(self.device)
        self.model_saver = ModelSaver(self.model_file)
        if self.config.get('parallel_computing') and _distributed() and dist.get_world_size() > 1:
            dp.broadcast_parameters(self.model)
            # the sparse word-row exchange (opt-in: UNITER_DP_SPARSE_EMB=1) is sized statically -- every caption is truncated and
            # padded to --max_txt_len (train_uniter.py:98, data/meme_dataset.py:170-177): batch_size x max_txt_len ids at most
            bs, tl = int(self.config.get('batch_size', 0) or 0), int(self.config.get('max_txt_len', 0) or 0)
            self.grad_sync = dp.attach(self.model, accum=int(self.config.get('gradient_accumulation', 1) or 1),
                                       token_capacity=(bs * tl) if (bs > 0 and tl > 0) else None)
            enc = getattr(self.model, 'uniter_model', None)
            if enc is not None:          # different dropout masks on every rank (the batches differ as well)
                enc.set_dropout_seed(int(self.config.get('seed', 0)) + 7919 * dist.get_rank())
        self.init_optimizer()
        self.init_scheduler()

    def init_scheduler(self):
        self.scheduler = get_scheduler(self.optimizer, self.config, len(self.config['train_loader']))

    def init_optimizer(self):
        self.optimizer = get_optimizer(self.model, self.config)
        enc = getattr(self.model, 'uniter_model', None)
        if enc is not None and self.config.get('overlap_optimizer', True) and hasattr(self.optimizer, 'overlap_encoder'):
            # the update of step i runs beside the forward of step i+1 (trainer.FusedAdam.step);
            # every other reader of the parameters joins first (ModelSaver.save)
            self.optimizer.overlap_encoder = enc
        if enc is not None and self.grad_sync is None and hasattr(self.optimizer, 'attach_norm_hooks') and \
                (self.config.get('max_grad_norm') or 0) > 0:
            self.optimizer.attach_norm_hooks(enc)    # clip norm reduced bucket by bucket during the backward pass
        if enc is not None and hasattr(self.optimizer, 'lazy_zero_encoder'):
            self.optimizer.lazy_zero_encoder = enc   # zero_grad skips the weight gradients the next backward pass overwrites

    # --------------------------------------------------------------------- step
    def _loss_and_probs(self, preds, labels):
        kind = self.config['loss_func']
        if kind == 'bce_logits':
            return bce_with_logits_loss(preds.squeeze(1), labels, self.config['pos_wt'], return_probs=True)
        if kind == 'bce':                # nn.BCELoss on torch.sigmoid(preds): the head emits logits (train_template.py:66-67,96-97)
            p = torch.sigmoid(preds.squeeze(1))
            return torch.nn.functional.binary_cross_entropy(p, labels.float()), p
        logp = torch.log_softmax(preds, dim=1)                                     # 'ce': two logits (:68-69)
        return torch.nn.functional.nll_loss(logp, labels.long()), logp[:, 1].exp()

    def calculate_loss(self, preds, batch_label, grad_step):
        """Step semantics of train_template.py:95-126.  The modulo test is on the per-epoch iteration
        index, so iteration 0 of every epoch steps with a single micro-batch that is still averaged
        over `gradient_accumulation` (kept: it is what the reference trains with)."""
        loss, probs = self._loss_and_probs(preds, batch_label)
        if grad_step:
            accum = self.config['gradient_accumulation']
            stepping = self.iters % accum == 0
            if self.grad_sync is not None:
                self.grad_sync.prepare(will_step=stepping, token_ids=getattr(self, 'batch', {}).get('input_ids'))
            loss.backward()
            if stepping:
                sync_step(self.optimizer, self.grad_sync, accum, self.config['max_grad_norm'])
                self.scheduler.step()
        self.log.add(probs, batch_label, loss, ids=self._ids)

    # --------------------------------------------------------------------- eval
    def _pass(self, loader, step):
        """One no-grad pass over `loader` into the epoch log; `step(iters, batch)` is the subclass hook."""
        self.model.eval()
        self.log.reset()
        wants_ids = bool(getattr(loader.dataset, 'return_ids', False))
        with torch.no_grad():
            for iters, batch in enumerate(loader):
                batch = self.batch_to_device(batch)
                self._ids = batch['ids'] if wants_ids else None
                step(iters, batch)
        self._ids = None

    def eval_model(self, test=False, test_idx=0):
        loader = self.config['test_loader'][test_idx] if test else self.config['val_loader']
        self._pass(loader, lambda iters, batch: self.eval_iter_step(iters, batch, test=test))
        self.eval_probs, self.eval_labels, loss, self.eval_ids = self.log.collect(with_ids=True)
        return standard_metrics(self.eval_probs, self.eval_labels, add_optimal_acc=True), loss

    def _stem(self):
        return os.path.join(self.config['model_path'], self.config['model_save_name'].rsplit('.', 1)[0])

    def _write_predictions(self, name, ids, probs, threshold, labels=None):
        with open('%s_%s_preds.csv' % (self._stem(), name), 'w') as f:
            f.write('id,proba,label' + (',gt' if labels is not None else '') + '\n')
            hard = (probs > threshold).long().tolist()
            for row, (i, p) in enumerate(zip(ids.tolist(), probs.tolist())):
                f.write('%i,%f,%i' % (i, p, hard[row]) + (',%i' % labels[row].item() if labels is not None else '') + '\n')

    @torch.no_grad()
    def export_test_predictions(self, test_idx=0, threshold=0.5):
        loader = self.config['test_loader'][test_idx]
        assert getattr(loader.dataset, 'return_ids', False), \
            "Can only export test results if the IDs are returned in the test dataset."
        self.model.eval()
        ids, probs = [], []
        for batch in loader:
            batch = self.batch_to_device(batch)
            ids.append(batch['ids'])
            probs.append(torch.sigmoid(self.test_iter_step(batch).reshape(-1)))
        self._write_predictions(loader.dataset.name, torch.cat(ids).cpu(), torch.cat(probs).cpu(), threshold)

    @torch.no_grad()
    def export_val_predictions(self, test=False, test_idx=0, threshold=0.5):
        loader = self.config['test_loader'][test_idx] if test else self.config['val_loader']
        self.eval_model(test=test, test_idx=test_idx)
        ids = self.eval_ids if self.eval_ids is not None else torch.full_like(self.eval_labels, -1)
        self._write_predictions(getattr(loader.dataset, 'name', 'val'), ids, self.eval_probs, threshold,
                                labels=self.eval_labels)

    def export_metrics(self):
        report = {'dev': dict(self.best_val_metrics, loss=self.best_val_loss),
                  'train': dict(self.train_metrics, loss=self.train_loss)}
        if self.test_metrics:
            report['test'] = self.test_metrics
        with open(self._stem() + '_metrics.json', 'w') as f:
            json.dump(report, f, indent=4)

    # ------------------------------------------------------------------- epochs
    def _log_metrics(self, group, metrics, step):
        w = self.config['writer']
        for tag, key in METRIC_TAGS:
            w.add_scalar('%s/%s' % (group, tag), metrics[key], step)

    def _train_epoch(self):
        self.log.reset()
        self._ids = None
        every = self.config['log_every']
        t0, seen = time.time(), 0
        for self.iters, self.batch in enumerate(self.config['train_loader']):
            self.model.train()
            self.batch = self.batch_to_device(self.batch)
            seen += int(self.batch['labels'].shape[0]) if torch.is_tensor(self.batch.get('labels')) else 0
            self.train_iter_step()
            done = self.total_iters + self.iters + 1
            if done % every == 0:
                w = self.config['writer']
                w.add_scalar('Train/Loss', self.log.recent_loss(), done)
                w.add_scalar('Stats/learning_rate', self.scheduler.get_last_lr()[0], done)
        self.total_iters += self.iters + 1
        probs, labels, self.train_loss = self.log.collect()        # (the epoch's one device -> host flush)
        self.train_rate = seen / max(time.time() - t0, 1e-9)       # samples per second of this rank, input pipeline included
        self.config['writer'].add_scalar('Stats/train_samples_per_s', self.train_rate, self.total_iters)
        self.train_metrics = standard_metrics(probs, labels, add_optimal_acc=True)

    def _after_epoch(self):
        lr = self.scheduler.get_last_lr()[0]
        w = self.config['writer']
        w.add_scalar('Train/Epoch_Loss', self.train_loss, self.total_iters)
        self._log_metrics('Train', self.train_metrics, self.epoch)
        w.add_scalar('Train/learning_rate', lr, self.epoch)
        t0 = time.time()
        self.val_metrics, self.val_loss = self.eval_model()
        w.add_scalar('Stats/time_validation', time.time() - t0, self.total_iters)
        w.add_scalar('Validation/Loss', self.val_loss, self.epoch)
        self._log_metrics('Validation', self.val_metrics, self.epoch)
        if _is_main():
            print("\nEpoch: {}/{},  train_loss = {:.4f}, train_acc = {:.4f}, train_aucroc = {:.4f}  |  "
                  "eval_loss = {:.4f}, eval_acc = {:.4f}, eval_aucroc = {:.4f}  |  lr = {:.8f}  elapsed {:.1f}s  "
                  "({:.0f} train samples/s)"
                  .format(self.epoch, self.config['max_epoch'], self.train_loss, self.train_metrics['accuracy'],
                          self.train_metrics['aucroc'], self.val_loss, self.val_metrics['accuracy'],
                          self.val_metrics['aucroc'], lr, time.time() - self.start, getattr(self, 'train_rate', 0.0)))
        key = self.config['optimize_for']
        improved, stop = self.plateau.update(self.val_loss if key == 'loss' else self.val_metrics[key])
        if _distributed() and dist.get_world_size() > 1:
            # every rank scored the validation set itself, and kernels that sum with float atomics differ between runs
            # in the last bits: a metric that sits on a threshold could let ONE rank stop (or not) and strand the others
            # in the next epoch's first collective.  Rank 0's verdict and plateau state are everybody's.
            verdict = torch.tensor([float(improved), float(stop), float(self.plateau.best), float(self.plateau.stale),
                                    float(self.val_loss)], dtype=torch.float64, device=self.device)
            dist.broadcast(verdict, src=0)
            v = verdict.tolist()
            improved, stop = bool(v[0]), bool(v[1])
            self.plateau.best, self.plateau.stale, self.val_loss = v[2], int(v[3]), v[4]
        if improved:
            LOGGER.info("New High Score! Saving model...")
            self.best_val_metrics, self.best_val_loss = self.val_metrics, self.val_loss
            if not self.config['no_model_checkpoints'] and _is_main():
                self.model_saver.save(self.model, self.optimizer)
        LOGGER.info("current patience: {}".format(self.plateau.stale))
        self.terminate_training = stop

    def _final_evaluation(self):
        """Reload the best checkpoint, pick the decision threshold on the validation set, score / export every test set."""
        if not os.path.isfile(self.model_file):
            raise ValueError("No Saved model state_dict found for the chosen model...!!! \n"
                             "Aborting evaluation on test set...")
        self.load_model()
        self.model.to(self.device)
        self.export_val_predictions()
        threshold = find_optimal_threshold(self.eval_probs, self.eval_labels, metric="accuracy")
        at_thr = standard_metrics(self.eval_probs, self.eval_labels, threshold=threshold, add_aucroc=False)
        LOGGER.info("Optimal threshold on validation dataset: %.4f (accuracy=%4.2f%%)"
                    % (threshold, 100.0 * at_thr["accuracy"]))
        for idx, loader in enumerate(self.config['test_loader']):
            name = getattr(loader.dataset, 'name', 'test%d' % idx)
            table = getattr(loader.dataset, 'data', None)
            if table is not None and hasattr(table, 'labels') and table.labels[0] == -1:     # unlabeled: predictions only
                self.export_test_predictions(test_idx=idx, threshold=threshold)
                self.test_metrics[name] = dict()
            else:
                self.test_metrics[name], _ = self.eval_model(test=True, test_idx=idx)
                self.export_val_predictions(test=True, test_idx=idx, threshold=threshold)

    def end_training(self):
        if self.terminate_training:
            LOGGER.info("Training terminated early because the Validation {} did not improve for   {}   epochs"
                        .format(self.config['optimize_for'], self.config['patience']))
        else:
            LOGGER.info("Maximum epochs of {} reached. Finished training !!".format(self.config['max_epoch']))
        self.test_metrics = dict()
        self._check_replicas()
        failure = None
        try:
            if self.config['no_model_checkpoints']:
                LOGGER.info("No model checkpoints were saved. Hence, testing will be skipped.")
            elif _is_main():
                self._final_evaluation()
            if _is_main():
                self.export_metrics()
                if self.config.get('remove_checkpoints') and os.path.isfile(self.model_file):
                    os.remove(self.model_file)
        except Exception as e:                   # noqa: BLE001 -- re-raised below, behind the barrier
            failure = e
        self.config['writer'].close()
        if _distributed():
            # rank 0 may still be scoring the test sets: nobody leaves (tears RCCL down, starts the next fold,
            # globs the prediction files) before the files exist -- and a rank 0 that failed still reaches the barrier,
            # it does not strand the others in it
            dist.barrier()
        if failure is not None:
            raise failure

    def _check_replicas(self):
        """Data parallel: every rank applied the same update to the same reduced gradients, so the replicas hold the same
        bits.  A difference means a gradient slice missed its collective -- fail loudly instead of exporting one rank's model."""
        if not (_distributed() and dist.get_world_size() > 1 and hasattr(self.model, 'param_store')):
            return
        flat = self.model.param_store().flat_params
        mine = torch.stack([flat.double().sum(), flat.double().abs().sum()])
        every = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        ref = every[0]
        for r, t in enumerate(every):
            if not torch.allclose(t, ref, rtol=1e-12, atol=0.0):
                raise RuntimeError('data-parallel replicas diverged: rank %d holds parameter sums %s, rank 0 %s'
                                   % (r, t.tolist(), ref.tolist()))
        LOGGER.info('data-parallel replicas agree (parameter checksum %.9e on %d ranks)' % (float(ref[1]), len(every)))

    def train_main(self, cache=False):
        self.start = time.time()
        if _is_main():
            print("\nBeginning training at:  {} \n".format(datetime.datetime.now()))
        self.model.to(self.device)
        for self.epoch in range(self.start_epoch, self.config['max_epoch'] + 1):
            self._train_epoch()
            self._after_epoch()
            if self.terminate_training:
                break
        self.end_training()
        return self.best_val_metrics, self.test_metrics

    def batch_to_device(self, batch):
        return {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) else v)
                for k, v in batch.items()}

    # ------------------------------------------------------------------- hooks
    def init_model(self):
        raise NotImplementedError

    def load_model(self):
        raise NotImplementedError

    def train_iter_step(self):
        raise NotImplementedError

    def eval_iter_step(self, iters, batch, test):
        raise NotImplementedError

    def test_iter_step(self, batch):
        raise NotImplementedError

    # --------------------------------------------------------------------- CLI
    # (flag, type or None for store_true, default) of train_template.py:424-506
    FLAGS = (('data_path', str, './dataset'), ('model_path', str, './model_checkpoints'),
             ('vis_path', str, './vis_checkpoints'), ('model_save_name', str, 'best_model.pt'),
             ('no_model_checkpoints', None, False), ('remove_checkpoints', None, False), ('debug', None, False),
             ('pretrained_model_file', str, None), ('optimizer', str, 'adam'), ('loss_func', str, 'bce_logits'),
             ('optimize_for', str, 'aucroc'), ('scheduler', str, 'warmup_cosine'), ('confounder_repeat', int, 1),
             ('object_conf_thresh', float, 0.0), ('num_folds', int, 0), ('crossval_dev_size', int, 300),
             ('crossval_use_dev', None, False), ('beta1', float, 0.9), ('beta2', float, 0.999),
             ('batch_size', int, 8), ('num_workers', int, 0), ('gradient_accumulation', int, 1),
             ('max_grad_norm', int, 5), ('pos_wt', float, 1), ('lr', float, 1e-4), ('warmup_steps', int, 50),
             ('weight_decay', float, 1e-3), ('max_epoch', int, 20), ('lr_decay_step', float, 3),
             ('lr_decay_factor', float, 0.8), ('patience', float, 5), ('early_stop_thresh', float, 1e-3),
             ('seed', int, 42), ('log_every', int, 2000), ('parallel_computing', bool, False))

    @classmethod
    def add_default_argparse(cls, parser, defaults=dict()):
        for name, kind, default in cls.FLAGS:
            if kind is None:
                parser.add_argument('--' + name, action='store_true')
            else:
                parser.add_argument('--' + name, type=kind, default=defaults.get(name, default))

    @staticmethod
    def preprocess_args(config, require_data_path=True):
        """What train_template.py:511-550 derives from the parsed flags: device, n_classes, the directories,
        the summary writer, the seed."""
        config['device'] = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
        config['n_classes'] = 2 if config['loss_func'] == 'ce' else 1
        if require_data_path and not os.path.exists(config['data_path']):
            raise ValueError("[!] ERROR: Dataset path does not exist")
        for key in ('model_path', 'vis_path'):
            os.makedirs(config[key], exist_ok=True)
        if 'config' in config:
            from .model import resolve_config
            resolve_config(config['config'])          # raises ValueError if neither a file nor a built-in size
        config['writer'] = _make_writer(config['vis_path'])
        set_seed(config['seed'])
        return config
