"""Input side of the hot path: the batch dict the model consumes.

Counterpart of data/dataset_template.py:92-114 (`_load_img_feature`: `<id>.npy` fp32
[nbb,2048] + `<id>_info.npy` pickled dict with bbox / image_width / image_height / objects /
objects_conf|cls_prob -> 7-d box features (x1,y1,x2,y2,w,h,w*h), normalised), of
data/meme_dataset.py:27-214 (jsonl list, `__getitem__`, `collate_fn` with compact batches) and
of `ConfounderSampler` (:221-271).  Same file formats, same batch keys:
  input_ids, position_ids, img_feat, img_pos_feat, token_type_ids, attn_mask, gather_index,
  labels, ids.
"""
import json
import os
from random import shuffle
from types import SimpleNamespace

import numpy as np
import torch
import torch.utils.data as data
from torch.nn.utils.rnn import pad_sequence

from .utils import get_attention_mask, get_gather_index


def expand_id(img_id):
    return str(img_id).zfill(5)


def load_img_feature(feature_dir, img_id, normalize=True):
    """-> (feat [nbb, D] fp32, pos [nbb, 7] fp32, objects, objects_conf)"""
    sid = expand_id(img_id)
    feat = torch.from_numpy(np.load(os.path.join(feature_dir, '%s.npy' % sid)))
    info = np.load(os.path.join(feature_dir, '%s_info.npy' % sid), allow_pickle=True).item()
    x1, y1, x2, y2 = [c.astype(np.float32).copy() for c in np.split(info['bbox'], 4, axis=1)]
    conf = info['objects_conf'] if 'objects_conf' in info else info['cls_prob'].max(axis=-1)
    if normalize:
        x1 /= info['image_width']; x2 /= info['image_width']
        y1 /= info['image_height']; y2 /= info['image_height']
    w, h = x2 - x1, y2 - y1
    pos = torch.from_numpy(np.concatenate((x1, y1, x2, y2, w, h, w * h), axis=1).astype(np.float32))
    return feat.float(), pos, info['objects'], conf


# ---- packed feature shard -------------------------------------------------------
# At >1000 samples/s per GPU the per-sample `<id>.npy` + pickled `<id>_info.npy` pair (two opens, one
# unpickle, 295 KB) is the input bottleneck.  A shard is the same data laid out for streaming: one
# fp32 matrix of all region features, one of the 7-d box features already computed as
# load_img_feature does, one of the detector confidences, and an index; all memory-mapped, so a
# sample is two slices of page cache and DataLoader workers share the pages.
SHARD_MAGIC = 'uniter-feature-shard-v1'


def build_feature_shard(feature_dir, img_ids, out_prefix, normalize=True):
    """Pack the reference-format feature files of `img_ids` into `<out_prefix>.{feat,pos,conf}.npy` +
    `<out_prefix>.index.json`.  Returns the index dict."""
    ids = [int(i) for i in img_ids]
    loaded = [load_img_feature(feature_dir, i, normalize=normalize) for i in ids]
    counts = [int(f.shape[0]) for f, _, _, _ in loaded]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    dim = int(loaded[0][0].shape[1]) if loaded else 0
    feat = np.lib.format.open_memmap(out_prefix + '.feat.npy', mode='w+', dtype=np.float32, shape=(int(offs[-1]), dim))
    pos = np.lib.format.open_memmap(out_prefix + '.pos.npy', mode='w+', dtype=np.float32, shape=(int(offs[-1]), 7))
    conf = np.lib.format.open_memmap(out_prefix + '.conf.npy', mode='w+', dtype=np.float32, shape=(int(offs[-1]),))
    for k, (f, p, _, c) in enumerate(loaded):
        feat[offs[k]:offs[k + 1]] = f.numpy()
        pos[offs[k]:offs[k + 1]] = p.numpy()
        conf[offs[k]:offs[k + 1]] = np.asarray(c, dtype=np.float32).reshape(-1)
    for a in (feat, pos, conf):
        a.flush()
    index = {'magic': SHARD_MAGIC, 'ids': ids, 'offsets': offs.tolist(), 'dim': dim, 'normalize': bool(normalize)}
    with open(out_prefix + '.index.json', 'w') as f:
        json.dump(index, f)
    return index


class FeatureShard(object):
    """Read side of build_feature_shard: `shard[img_id] -> (feat [nbb, D], pos [nbb, 7], conf [nbb])` as torch
    views of the memory map (zero copy until collate pads them)."""

    def __init__(self, prefix):
        with open(prefix + '.index.json') as f:
            idx = json.load(f)
        if idx.get('magic') != SHARD_MAGIC:
            raise ValueError('%s.index.json is not a feature shard' % prefix)
        self.offsets = np.asarray(idx['offsets'], dtype=np.int64)
        self.row = {int(i): k for k, i in enumerate(idx['ids'])}
        self.feat = np.load(prefix + '.feat.npy', mmap_mode='r')
        self.pos = np.load(prefix + '.pos.npy', mmap_mode='r')
        self.conf = np.load(prefix + '.conf.npy', mmap_mode='r')

    def __contains__(self, img_id):
        return int(img_id) in self.row

    def __getitem__(self, img_id):
        k = self.row[int(img_id)]
        a, b = int(self.offsets[k]), int(self.offsets[k + 1])
        # np.array(copy) of a contiguous slice: one memcpy out of the page cache, then an owned tensor
        return (torch.from_numpy(np.array(self.feat[a:b])), torch.from_numpy(np.array(self.pos[a:b])),
                np.array(self.conf[a:b]))


class DevicePrefetcher(object):
    """Iterates a DataLoader one batch ahead: batch i+1 is copied to the GPU on a side stream (from the
    loader's pinned buffers, non-blocking) while step i computes, so the hot path never waits for PCIe.
    Non-tensor entries (seq_lens lists, None) pass through."""

    def __init__(self, loader, device, threaded=None, depth=3):
        """threaded (round 6; None: on unless UNITER_PREFETCH_THREAD=0): the loader's own work -- reading the feature files and the
        collate of data/meme_dataset.py:152-214, 2.3 ms per batch of configs[1] with --num_workers 0 -- runs in a background thread, up
        to `depth` batches ahead, instead of between two steps on the thread that launches the kernels: with a 4.4-ms bf16 step the
        launching thread had 2 ms of it taken away every iteration (the CLI ran at 72-80 % of the bare step's rate)."""
        self.loader, self.device = loader, torch.device(device)
        from . import _lib
        self.stream = _lib.shared_stream(self.device, 'copy')      # one copy stream per device, whatever the number of loaders
        self.threaded = (os.environ.get('UNITER_PREFETCH_THREAD', '1') != '0') if threaded is None else bool(threaded)
        self.depth = max(1, int(depth))

    def __len__(self):
        return len(self.loader)

    @property
    def dataset(self):
        return self.loader.dataset

    def _to_device(self, batch):
        with torch.cuda.stream(self.stream):
            return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    def _iter_threaded(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        # (the consuming thread's device: a new thread starts on device 0 whatever the caller selected)
        dev_index = (self.device.index if self.device.index is not None else torch.cuda.current_device()) if self.device.type == 'cuda' else None

        def work():
            try:
                if dev_index is not None:
                    torch.cuda.set_device(dev_index)
                for b in self.loader:
                    d = self._to_device(b)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                    if not put((d, ev)):
                        return
                put(None)
            except BaseException as e:             # noqa: BLE001 -- handed to the consuming thread, which raises it
                put(e)

        t = threading.Thread(target=work, name='uniter-prefetch', daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                cur, ev = item
                cs = torch.cuda.current_stream(self.device)
                cs.wait_event(ev)                                                 # cur's copies are done
                for v in cur.values():
                    if torch.is_tensor(v):
                        v.record_stream(cs)                                       # allocator: in use on the compute stream
                yield cur
        finally:
            stop.set()
            while not q.empty():
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            t.join(timeout=10)

    def __iter__(self):
        if self.threaded:
            yield from self._iter_threaded()
            return
        it = iter(self.loader)
        try:
            nxt = self._to_device(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            torch.cuda.current_stream(self.device).wait_stream(self.stream)       # cur's copies are done
            for v in cur.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream(self.device))       # allocator: in use on the compute stream
            try:
                nxt = self._to_device(next(it))                                  # overlaps with the caller's step on `cur`
            except StopIteration:
                nxt = None
            yield cur


class MemeDataset(data.Dataset):
    def __init__(self, filepath, feature_dir=None, text_padding=None, return_ids=False, compact_batch=True,
                 confidence_threshold=0.0, preload_images=False, text_only=False, debug=False,
                 feature_shard=None, ragged_regions=False, **unused):
        """`ragged_regions=False` reproduces the reference's collate exactly: it measures the region count of every
        sample on the already zero-padded stack (data/meme_dataset.py:161,191), so every sample of a batch carries the
        batch's largest region count and the zero rows are attended like real regions (pinned by
        tests/golden/data_pipeline.npz).  True masks each sample at its own region count instead (what
        utils/utils.py:111-125 describes; fewer valid positions for token packing)."""
        self.ragged_regions = ragged_regions
        assert os.path.isfile(filepath), "Dataset file cannot be found: \"%s\"." % filepath
        assert filepath.endswith(".jsonl"), "The filepath requires a JSON list file (\".jsonl\")."
        self.filepath, self.feature_dir = filepath, feature_dir
        self.name = filepath.split("/")[-1].split(".")[0]
        self.text_padding, self.return_ids, self.compact_batch = text_padding, return_ids, compact_batch
        self.confidence_threshold, self.text_only = confidence_threshold, text_only
        with open(filepath, "r") as f:
            js = [json.loads(l) for l in f if l.strip()]
        self.data = SimpleNamespace(
            ids=torch.LongTensor([int(j["id"]) for j in js]),
            labels=torch.LongTensor([j.get("label", -1) for j in js]),
            text=[j["text"] for j in js],
            imgs=[os.path.join(os.path.dirname(filepath), j.get("img", "")) for j in js])
        self.shard = FeatureShard(feature_shard) if (feature_shard and not text_only) else None
        if not text_only:
            for i in self.data.ids.tolist():
                if self.shard is not None:
                    assert i in self.shard, "Image %d is not in the feature shard %s." % (i, feature_shard)
                    continue
                for suffix in ('.npy', '_info.npy'):
                    p = os.path.join(feature_dir, expand_id(i) + suffix)
                    assert os.path.isfile(p), "Feature file %s does not exist." % p
        self._cache = None
        if preload_images and not text_only:
            self._cache = [load_img_feature(feature_dir, i) for i in self.data.ids.tolist()]

    def __len__(self):
        return len(self.data.ids)

    def __getitem__(self, idx):
        data_id, label = self.data.ids[idx], self.data.labels[idx]
        feat = pos = None
        if not self.text_only:
            if self._cache is not None:
                feat, pos, _, conf = self._cache[idx]
            elif self.shard is not None:
                feat, pos, conf = self.shard[data_id.item()]
            else:
                feat, pos, _, conf = load_img_feature(self.feature_dir, data_id.item())
            if self.confidence_threshold > 0.0:
                keep = torch.from_numpy(np.asarray(conf) > self.confidence_threshold)
                feat, pos = feat[keep], pos[keep]
        return {'img_feat': feat, 'img_pos_feat': pos, 'text': self.data.text[idx], 'label': label,
                'data_id': data_id}

    def get_collate_fn(self):
        def collate_fn(samples):
            texts = self.text_padding([s['text'] for s in samples])
            input_ids = texts['input_ids']
            text_len = [int(x) for x in (texts['length'].tolist() if torch.is_tensor(texts['length'])
                                         else texts['length'])]
            B, T = input_ids.shape
            batch = {'input_ids': input_ids,
                     'position_ids': torch.arange(T, dtype=torch.long).unsqueeze(0).repeat(B, 1),
                     'token_type_ids': texts.get('token_type_ids') if hasattr(texts, 'get') else None,
                     'labels': torch.stack([s['label'] for s in samples]),
                     'ids': torch.stack([s['data_id'] for s in samples])}
            if self.text_only:
                batch.update(img_feat=None, img_pos_feat=None, attn_mask=texts['attention_mask'], gather_index=None)
                return batch
            img_feat = pad_sequence([s['img_feat'] for s in samples], batch_first=True, padding_value=0)
            img_pos = pad_sequence([s['img_pos_feat'] for s in samples], batch_first=True, padding_value=0)
            img_len = [s['img_feat'].size(0) for s in samples] if self.ragged_regions else [img_feat.size(1)] * B
            if self.compact_batch:
                attn = get_attention_mask(text_len, img_len)
            else:
                attn = torch.cat((texts['attention_mask'].float(), get_attention_mask([0] * B, img_len)), dim=1)
            gi = get_gather_index(text_len, img_len, B, T, attn.shape[1])
            batch.update(img_feat=img_feat, img_pos_feat=img_pos, attn_mask=attn, gather_index=gi)
            if self.compact_batch:
                # host-side lengths (extension): lets UniterModel.pack_padded skip a device->host copy
                batch['seq_lens'] = [int(t) + int(n) for t, n in zip(text_len, img_len)]
            return batch
        return collate_fn


class ConfounderSampler(data.Sampler):
    """Repeats the text confounders (same text, both labels) `repeat_factor` times per epoch
    (data/meme_dataset.py:221-271)."""

    def __init__(self, dataset, repeat_factor=1):
        self.dataset, self.repeat_factor = dataset, repeat_factor
        labels_of = {}
        for idx, text in enumerate(dataset.data.text):
            labels_of.setdefault(text, set()).add(int(dataset.data.labels[idx]))
        conf_text = {t for t, ls in labels_of.items() if ls == {0, 1}}
        self.confounders = [i for i, t in enumerate(dataset.data.text) if t in conf_text]
        self.non_confounders = [i for i, t in enumerate(dataset.data.text) if t not in conf_text]
        self._generate()

    def _generate(self):
        plain = self.non_confounders[:]
        shuffle(plain)
        n, k = len(plain), self.repeat_factor
        splits = [(n // k) * i for i in range(k)] + [n]
        out = []
        for i in range(k):
            sub = plain[splits[i]:splits[i + 1]] + self.confounders
            shuffle(sub)
            out += sub
        self.sample_list = out

    def __iter__(self):
        self._generate()
        return iter(self.sample_list)

    def __len__(self):
        return len(self.sample_list)


class HashTokenizer(object):
    """Offline stand-in for `BertTokenizer('bert-base-cased')` (which needs the hub): lower-cased
    whitespace tokens hashed into the vocabulary, [CLS]=101 / [SEP]=102 / [PAD]=0, same call
    signature and return fields as the partial built at train_uniter.py:124-126."""

    def __init__(self, vocab_size=28996, max_length=60):
        self.vocab_size, self.max_length = vocab_size, max_length

    def __call__(self, texts, **kw):
        T = kw.get('max_length', self.max_length)
        ids = torch.zeros(len(texts), T, dtype=torch.long)
        lens = []
        for b, t in enumerate(texts):
            toks = [101] + [1000 + (hash_str(w) % (self.vocab_size - 1000)) for w in str(t).lower().split()][:T - 2] + [102]
            ids[b, :len(toks)] = torch.tensor(toks)
            lens.append(len(toks))
        return {'input_ids': ids, 'length': torch.tensor(lens), 'attention_mask': (ids != 0).long(),
                'token_type_ids': torch.zeros_like(ids)}


def hash_str(s):
    h = 2166136261
    for c in s.encode('utf-8'):
        h = ((h ^ c) * 16777619) & 0xFFFFFFFF
    return h


def write_synthetic_dataset(root, n=64, num_bb=(10, 36), img_dim=2048, seed=0, splits=('train', 'dev_seen'), text_words=(3, 8)):
    """Create a dataset in the reference's ON-DISK FORMAT (jsonl + per-image .npy feature files) with
    a learnable signal: label 1 memes have a shifted feature mean and contain a marker word.  text_words / num_bb: (min, max)
    words per caption and regions per image (BASELINE configs[1] shapes: text_words=(126, 126), num_bb=(36, 36))."""
    rng = np.random.default_rng(seed)
    feat_dir = os.path.join(root, 'img_feats')
    os.makedirs(feat_dir, exist_ok=True)
    words = ['cat', 'dog', 'meme', 'funny', 'sky', 'tree', 'car', 'look', 'when', 'you', 'me', 'nobody']
    idx = 0
    for split in splits:
        with open(os.path.join(root, split + '.jsonl'), 'w') as f:
            for _ in range(n):
                label = int(rng.random() < 0.4)
                nbb = int(rng.integers(num_bb[0], num_bb[1] + 1))
                feat = np.abs(rng.standard_normal((nbb, img_dim))).astype(np.float32) + 0.5 * label
                W, H = 640, 480
                x1 = rng.random((nbb, 1)) * 0.7 * W; y1 = rng.random((nbb, 1)) * 0.7 * H
                bw = (rng.random((nbb, 1)) * 0.25 + 0.05) * W; bh = (rng.random((nbb, 1)) * 0.25 + 0.05) * H
                info = {'bbox': np.concatenate([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32),
                        'image_width': W, 'image_height': H,
                        'objects': rng.integers(0, 1600, nbb), 'objects_conf': rng.random(nbb).astype(np.float32)}
                np.save(os.path.join(feat_dir, expand_id(idx) + '.npy'), feat)
                np.save(os.path.join(feat_dir, expand_id(idx) + '_info.npy'), info, allow_pickle=True)
                text = ' '.join(rng.choice(words, int(rng.integers(text_words[0], text_words[1] + 1)))) + (' hateful' if label else ' nice')
                f.write(json.dumps({'id': idx, 'img': 'img/%s.png' % expand_id(idx), 'label': label, 'text': text}) + '\n')
                idx += 1
    return feat_dir
