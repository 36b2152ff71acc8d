"""Trainer scaffolding around the HIP hot path: counterpart of the reference's
`TrainerTemplate` (train_template.py:26-552) with the same hook contract
(`init_model / load_model / train_iter_step / eval_iter_step / test_iter_step`),
the same CLI flags (`add_default_argparse`), the same step semantics
(`calculate_loss`), early stopping on the same metrics, the same checkpoint
format (`{'model_state_dict': ...}`, utils/save.py:57-63) and the same
prediction / metric exports (`id,proba,label[,gt]` CSV, `_metrics.json`).

Differences, all on purpose:
  * per-iteration results (loss, probabilities, labels) stay on the GPU and are flushed
    once per epoch -- the reference synchronises with `.item()`/`.cpu()` every iteration
    (train_template.py:121-124);
  * averaging, clipping, the optimizer update and zero_grad are one fused kernel;
  * `--parallel_computing` selects one-process-per-GPU data parallelism over RCCL (launch
    with torchrun) instead of single-process nn.DataParallel;
  * tensorboard is optional (scalars are logged only when it is importable).
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
from .trainer import bce_with_logits_loss, get_optimizer, get_scheduler
from .utils import set_seed

LOGGER = logging.getLogger('TrainerLogger')
logging.basicConfig(format='%(asctime)s : %(levelname)s - %(message)s', datefmt='%d/%m/%Y %I:%M:%S %p',
                    level=logging.INFO)


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
    """utils/save.py:53-63: torch.save({'model_state_dict': cpu tensors})."""

    def __init__(self, output_path):
        self.output_path = output_path

    def save(self, model, optimizer=None):
        if optimizer is not None and hasattr(optimizer, 'join'):
            optimizer.join()
        state_dict = {k: v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v
                      for k, v in model.state_dict().items()}
        torch.save({'model_state_dict': state_dict}, self.output_path)


class TrainerTemplate(object):

    def __init__(self, config):
        self.probs_list, self.labels_list, self.loss_list, self.short_loss_list, self.id_list = [], [], [], [], []
        self.best_val_metrics, self.train_metrics = defaultdict(int), {}
        self.not_improved = 0
        self.best_val_loss = 1000
        self.total_iters = 0
        self.terminate_training = False
        self.model_file = os.path.join(config['model_path'], config['model_save_name'])
        self.pretrained_model_file = None
        if config.get('pretrained_model_file') is not None:
            self.pretrained_model_file = os.path.join(config['model_path'], config['pretrained_model_file'])
        self.start_epoch = 1
        self.config = config
        self.device = config.get('device', torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))))
        if not isinstance(self.config['test_loader'], list):
            self.config['test_loader'] = [self.config['test_loader']]
        self.grad_sync = None
        self.init_training_params()

    # ------------------------------------------------------------------ set-up
    def init_training_params(self):
        self.init_model()
        self.model.to(self.device)
        self.model_saver = ModelSaver(self.model_file)
        if self.config.get('parallel_computing') and dist.is_available() and dist.is_initialized() \
                and dist.get_world_size() > 1:
            dp.broadcast_parameters(self.model)
            self.grad_sync = dp.attach(self.model)
        self.init_optimizer()
        self.init_scheduler()
        if self.config['loss_func'] != 'bce_logits':
            raise ValueError("loss_func=%r: only 'bce_logits' is built on the HIP path" % self.config['loss_func'])

    def init_scheduler(self):
        self.scheduler = get_scheduler(self.optimizer, self.config, len(self.config['train_loader']))

    def init_optimizer(self):
        self.optimizer = get_optimizer(self.model, self.config)
        enc = getattr(self.model, 'uniter_model', None)
        if enc is not None and self.config.get('overlap_optimizer', True):
            # the update of step i runs beside the forward of step i+1 (trainer.FusedAdam.step);
            # every other reader of the parameters joins first (ModelSaver.save below)
            self.optimizer.overlap_encoder = enc

    # --------------------------------------------------------------------- step
    def calculate_loss(self, preds, batch_label, grad_step):
        """train_template.py:95-126.  The modulo test is on the per-epoch iteration index
        (iteration 0 of every epoch steps with one micro-batch, still averaged over
        `gradient_accumulation`)."""
        cfg = self.config
        loss, probs = bce_with_logits_loss(preds.squeeze(1), batch_label, cfg['pos_wt'], return_probs=True)
        if grad_step:
            accum = cfg['gradient_accumulation']
            stepping = self.iters % accum == 0
            if self.grad_sync is not None:
                self.grad_sync.prepare(will_step=stepping)
            loss.backward()
            if stepping:
                world = 1
                if self.grad_sync is not None:
                    self.grad_sync.finish()
                    world = self.grad_sync.world
                self.optimizer.step(grad_scale=1.0 / (accum * world), max_grad_norm=cfg['max_grad_norm'],
                                    zero_grads=True)
                self.scheduler.step()
        # no host synchronisation here: flushed in _flush_epoch_lists()
        self.probs_list.append(probs.detach())
        self.labels_list.append(batch_label.detach())
        self.loss_list.append(loss.detach())
        if grad_step:
            self.short_loss_list.append(loss.detach())

    def _flush_epoch_lists(self):
        probs = torch.cat(self.probs_list).float().cpu() if self.probs_list else torch.zeros(0)
        labels = torch.cat(self.labels_list).cpu() if self.labels_list else torch.zeros(0, dtype=torch.long)
        losses = torch.stack(self.loss_list).float().cpu().tolist() if self.loss_list else []
        return probs, labels, losses

    # --------------------------------------------------------------------- eval
    def eval_model(self, test=False, test_idx=0):
        self.model.eval()
        self.probs_list, self.labels_list, self.loss_list, self.id_list = [], [], [], []
        loader = self.config['val_loader'] if not test else self.config['test_loader'][test_idx]
        with torch.no_grad():
            for iters, batch in enumerate(loader):
                batch = self.batch_to_device(batch)
                if getattr(loader.dataset, 'return_ids', False):
                    self.id_list.append(batch['ids'])
                self.eval_iter_step(iters, batch, test=test)
        self.eval_probs, self.eval_labels, losses = self._flush_epoch_lists()
        self.eval_ids = torch.cat(self.id_list).cpu() if self.id_list else None
        val_loss = sum(losses) / max(len(losses), 1)
        return standard_metrics(self.eval_probs, self.eval_labels, add_optimal_acc=True), val_loss

    @torch.no_grad()
    def export_test_predictions(self, test_idx=0, threshold=0.5):
        self.model.eval()
        loader = self.config['test_loader'][test_idx]
        assert getattr(loader.dataset, 'return_ids', False), \
            "Can only export test results if the IDs are returned in the test dataset."
        prob_list, id_list = [], []
        for batch in loader:
            batch = self.batch_to_device(batch)
            id_list.append(batch['ids'])
            prob_list.append(torch.sigmoid(self.test_iter_step(batch).reshape(-1)))
        probs = torch.cat(prob_list).cpu()
        ids = torch.cat(id_list).cpu()
        self._export_preds(ids, probs, (probs > threshold).long(), file_postfix="_%s_preds.csv" % loader.dataset.name)

    @torch.no_grad()
    def export_val_predictions(self, test=False, test_idx=0, threshold=0.5):
        loader = self.config['val_loader'] if not test else self.config['test_loader'][test_idx]
        self.eval_model(test=test, test_idx=test_idx)
        ids = self.eval_ids if self.eval_ids is not None else torch.zeros_like(self.eval_labels) - 1
        self._export_preds(ids, self.eval_probs, (self.eval_probs > threshold).long(), labels=self.eval_labels,
                           file_postfix="_%s_preds.csv" % getattr(loader.dataset, 'name', 'val'))

    def _export_preds(self, ids, probs, preds, labels=None, file_postfix="_preds.csv"):
        lines = ["id,proba,label%s" % (",gt" if labels is not None else "")]
        for i in range(ids.shape[0]):
            row = "%i,%f,%i" % (ids[i].item(), probs[i].item(), preds[i].item())
            if labels is not None:
                row += ",%i" % labels[i].item()
            lines.append(row)
        path = os.path.join(self.config['model_path'], self.config['model_save_name'].rsplit(".", 1)[0] + file_postfix)
        with open(path, "w") as f:
            f.write("\n".join(lines) + "\n")

    # ---------------------------------------------------------- early stopping
    def check_early_stopping(self):
        key = self.config['optimize_for']
        this = self.val_loss if key == 'loss' else self.val_metrics[key]
        best = self.best_val_loss if key == 'loss' else self.best_val_metrics[key]
        new_best = this < best if key == 'loss' else this > best
        if new_best:
            LOGGER.info("New High Score! Saving model...")
            self.best_val_metrics = self.val_metrics
            self.best_val_loss = self.val_loss
            if not self.config["no_model_checkpoints"] and self._is_main():
                self.model_saver.save(self.model, self.optimizer)
        diff = best - this if key == 'loss' else this - best
        if diff < self.config['early_stop_thresh']:
            self.not_improved += 1
            if self.not_improved >= self.config['patience']:
                self.terminate_training = True
        else:
            self.not_improved = 0
        LOGGER.info("current patience: {}".format(self.not_improved))

    @staticmethod
    def _is_main():
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    # ------------------------------------------------------------------- epochs
    def train_epoch_step(self):
        lr = self.scheduler.get_last_lr()
        self.total_iters += self.iters + 1
        probs, labels, losses = self._flush_epoch_lists()
        self.train_metrics = standard_metrics(probs, labels, add_optimal_acc=True)
        self.train_loss = sum(losses) / max(len(losses), 1)
        w = self.config['writer']
        w.add_scalar('Train/Epoch_Loss', self.train_loss, self.total_iters)
        for k_tb, k in (('F1', 'F1'), ('Precision', 'precision'), ('Recall', 'recall'), ('Accuracy', 'accuracy'),
                        ('AUC-ROC', 'aucroc')):
            w.add_scalar('Train/' + k_tb, self.train_metrics[k], self.epoch)
        w.add_scalar("Train/learning_rate", lr[0], self.epoch)
        t0 = time.time()
        self.val_metrics, self.val_loss = self.eval_model()
        w.add_scalar("Stats/time_validation", time.time() - t0, self.total_iters)
        w.add_scalar('Validation/Loss', self.val_loss, self.epoch)
        for k_tb, k in (('F1', 'F1'), ('Precision', 'precision'), ('Recall', 'recall'), ('Accuracy', 'accuracy'),
                        ('AUC-ROC', 'aucroc')):
            w.add_scalar('Validation/' + k_tb, self.val_metrics[k], self.epoch)
        if self._is_main():
            print("\nEpoch: {}/{},  train_loss = {:.4f}, train_acc = {:.4f}, train_aucroc = {:.4f}  |  "
                  "eval_loss = {:.4f}, eval_acc = {:.4f}, eval_aucroc = {:.4f}  |  lr = {:.8f}  elapsed {:.1f}s"
                  .format(self.epoch, self.config['max_epoch'], self.train_loss, self.train_metrics['accuracy'],
                          self.train_metrics['aucroc'], self.val_loss, self.val_metrics['accuracy'],
                          self.val_metrics['aucroc'], lr[0], time.time() - self.start))
        self.check_early_stopping()
        self.probs_list, self.labels_list, self.loss_list, self.id_list = [], [], [], []

    def end_training(self):
        if self.terminate_training:
            LOGGER.info("Training terminated early because the Validation {} did not improve for   {}   epochs"
                        .format(self.config['optimize_for'], self.config['patience']))
        else:
            LOGGER.info("Maximum epochs of {} reached. Finished training !!".format(self.config['max_epoch']))
        self.test_metrics = dict()
        if not self.config["no_model_checkpoints"] and self._is_main():
            if not os.path.isfile(self.model_file):
                raise ValueError("No Saved model state_dict found for the chosen model...!!! \n"
                                 "Aborting evaluation on test set...")
            self.load_model()
            self.model.to(self.device)
            self.export_val_predictions()
            threshold = find_optimal_threshold(self.eval_probs, self.eval_labels, metric="accuracy")
            best = standard_metrics(self.eval_probs, self.eval_labels, threshold=threshold, add_aucroc=False)
            LOGGER.info("Optimal threshold on validation dataset: %.4f (accuracy=%4.2f%%)"
                        % (threshold, 100.0 * best["accuracy"]))
            for test_idx, loader in enumerate(self.config['test_loader']):
                name = getattr(loader.dataset, 'name', 'test%d' % test_idx)
                data = getattr(loader.dataset, 'data', None)
                unlabeled = data is not None and hasattr(data, 'labels') and data.labels[0] == -1
                if unlabeled:
                    self.export_test_predictions(test_idx=test_idx, threshold=threshold)
                    self.test_metrics[name] = dict()
                else:
                    m, _ = self.eval_model(test=True, test_idx=test_idx)
                    self.test_metrics[name] = m
                    self.export_val_predictions(test=True, test_idx=test_idx, threshold=threshold)
        else:
            LOGGER.info("No model checkpoints were saved. Hence, testing will be skipped.")
        if self._is_main():
            self.export_metrics()
        self.config['writer'].close()
        if self.config.get('remove_checkpoints') and self._is_main() and os.path.isfile(self.model_file):
            os.remove(self.model_file)

    def export_metrics(self):
        path = os.path.join(self.config['model_path'], self.config['model_save_name'].rsplit(".", 1)[0] + "_metrics.json")
        d = {"dev": dict(self.best_val_metrics), "train": dict(self.train_metrics)}
        d["dev"]["loss"] = self.best_val_loss
        d["train"]["loss"] = self.train_loss
        if getattr(self, "test_metrics", None):
            d["test"] = self.test_metrics
        with open(path, "w") as f:
            json.dump(d, f, indent=4)

    def train_main(self, cache=False):
        self.start = time.time()
        if self._is_main():
            print("\nBeginning training at:  {} \n".format(datetime.datetime.now()))
        self.model.to(self.device)
        for self.epoch in range(self.start_epoch, self.config['max_epoch'] + 1):
            for self.iters, self.batch in enumerate(self.config['train_loader']):
                self.model.train()
                self.batch = self.batch_to_device(self.batch)
                self.train_iter_step()
                if (self.total_iters + self.iters + 1) % self.config['log_every'] == 0:
                    sl = torch.stack(self.short_loss_list).mean().item()
                    self.config['writer'].add_scalar('Train/Loss', sl, self.iters + 1 + self.total_iters)
                    self.config['writer'].add_scalar('Stats/learning_rate', self.scheduler.get_last_lr()[0],
                                                     self.iters + self.total_iters + 1)
                    self.short_loss_list = []
            self.train_epoch_step()
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
    @staticmethod
    def add_default_argparse(parser, defaults=dict()):
        """The flags of train_template.py:424-506, same names / types / defaults."""
        g = defaults.get
        parser.add_argument('--data_path', type=str, default='./dataset')
        parser.add_argument('--model_path', type=str, default='./model_checkpoints')
        parser.add_argument('--vis_path', type=str, default='./vis_checkpoints')
        parser.add_argument("--model_save_name", type=str, default='best_model.pt')
        parser.add_argument("--no_model_checkpoints", action="store_true")
        parser.add_argument("--remove_checkpoints", action="store_true")
        parser.add_argument('--debug', action="store_true")
        parser.add_argument('--pretrained_model_file', type=str)
        parser.add_argument('--optimizer', type=str, default=g('optimizer', 'adam'))
        parser.add_argument('--loss_func', type=str, default=g('loss_func', 'bce_logits'))
        parser.add_argument('--optimize_for', type=str, default=g('optimize_for', 'aucroc'))
        parser.add_argument('--scheduler', type=str, default=g('scheduler', 'warmup_cosine'))
        parser.add_argument('--confounder_repeat', type=int, default=g('confounder_repeat', 1))
        parser.add_argument('--object_conf_thresh', type=float, default=g('object_conf_thresh', 0.0))
        parser.add_argument('--num_folds', type=int, default=g('num_folds', 0))
        parser.add_argument('--crossval_dev_size', type=int, default=g('crossval_dev_size', 300))
        parser.add_argument('--crossval_use_dev', action="store_true")
        parser.add_argument('--beta1', type=float, default=g('beta1', 0.9))
        parser.add_argument('--beta2', type=float, default=g('beta2', 0.999))
        parser.add_argument('--batch_size', type=int, default=g('batch_size', 8))
        parser.add_argument('--num_workers', type=int, default=g('num_workers', 0))
        parser.add_argument('--gradient_accumulation', type=int, default=g('gradient_accumulation', 1))
        parser.add_argument('--max_grad_norm', type=int, default=g('max_grad_norm', 5))
        parser.add_argument('--pos_wt', type=float, default=g('pos_wt', 1))
        parser.add_argument('--lr', type=float, default=g('lr', 1e-4))
        parser.add_argument('--warmup_steps', type=int, default=g('warmup_steps', 50))
        parser.add_argument('--weight_decay', type=float, default=g('weight_decay', 1e-3))
        parser.add_argument('--max_epoch', type=int, default=g('max_epoch', 20))
        parser.add_argument('--lr_decay_step', type=float, default=g('lr_decay_step', 3))
        parser.add_argument('--lr_decay_factor', type=float, default=g('lr_decay_factor', 0.8))
        parser.add_argument('--patience', type=float, default=g('patience', 5))
        parser.add_argument('--early_stop_thresh', type=float, default=g('early_stop_thresh', 1e-3))
        parser.add_argument('--seed', type=int, default=g('seed', 42))
        parser.add_argument('--log_every', type=int, default=g('log_every', 2000))
        parser.add_argument('--parallel_computing', type=bool, default=g('parallel_computing', False))

    @staticmethod
    def preprocess_args(config, require_data_path=True):
        """train_template.py:511-550: path checks, n_classes, writer, seed."""
        config['device'] = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
        config['n_classes'] = 2 if config['loss_func'] == 'ce' else 1
        if require_data_path and not os.path.exists(config['data_path']):
            raise ValueError("[!] ERROR: Dataset path does not exist")
        os.makedirs(config['model_path'], exist_ok=True)
        if 'config' in config:
            from .model import resolve_config
            resolve_config(config['config'])          # raises ValueError if neither a file nor a built-in size
        os.makedirs(config['vis_path'], exist_ok=True)
        config['writer'] = _make_writer(config['vis_path'])
        set_seed(config['seed'])
        return config
