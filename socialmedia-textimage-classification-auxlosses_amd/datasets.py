"""Input pipeline to the model boundary -- mirrors reference models/datasets.py:125-190 (MM_Dataset) and
models/utils.py:133-200 (prepare_data): tweet normalisation, tokenisation to max_length=128 with padding='max_length',
image -> 224x224, rescale 1/255, mean = std = 0.5 (the ViT image processor's defaults), one-hot int64 labels."""
import logging
import os

import numpy as np
import torch

from .config import MODEL_DIR_DICT
from .utils import balanced_class_weights

logger = logging.getLogger(__name__)

_SPECIAL_PUNCTS = {"\u2019": "'", "\u2026": "..."}
_CONTRACTIONS = (("cannot ", "can not "), ("n't ", " n't "), ("n 't ", " n't "), ("ca n't", "can't"), ("ai n't", "ain't"))
_CLITICS = (("'m ", " 'm "), ("'re ", " 're "), ("'s ", " 's "), ("'ll ", " 'll "), ("'d ", " 'd "), ("'ve ", " 've "))
_CLOCK = ((" p . m .", "  p.m."), (" p . m ", " p.m "), (" a . m .", " a.m."), (" a . m ", " a.m "))
_tokenizer = None
_demojize = False


def _tweet_tokenizer():
    """the NLTK-derived tweet tokenizer the reference vendors (text_processing.py:250-): transformers ships the same class
    with BERTweet (tokenization_bertweet.TweetTokenizer); equality of the two is part of tests/golden/text_golden.json"""
    global _tokenizer, _demojize
    if _tokenizer is None:
        from transformers.models.bertweet.tokenization_bertweet import TweetTokenizer
        _tokenizer = TweetTokenizer()
        try:
            from emoji import demojize                      # optional in the reference too (text_processing.py:186-196)
            _demojize = demojize
        except ImportError:
            _demojize = None
    return _tokenizer


def _normalize_token(token):
    """reference text_processing.py:231-248"""
    low = token.lower()
    if token.startswith("@"):
        return "@USER"
    if low.startswith("http") or low.startswith("www"):
        return "HTTPURL"
    if len(token) == 1:
        if token in _SPECIAL_PUNCTS:
            return _SPECIAL_PUNCTS[token]
        return _demojize(token) if _demojize else token
    return token


def normalize_tweet(text):
    """reference preprocessing/text_processing.py:198-228 (`Tweet_Preprocessing.normalizeTweet`): typographic quote /
    ellipsis folding, tweet tokenisation, @USER / HTTPURL, contraction and clock-time re-spacing, whitespace squeeze"""
    text = str(text)
    for k, v in _SPECIAL_PUNCTS.items():
        text = text.replace(k, v)
    out = " ".join(_normalize_token(t) for t in _tweet_tokenizer().tokenize(text))
    for table in (_CONTRACTIONS, _CLITICS, _CLOCK):
        for a, b in table:
            out = out.replace(a, b)
    return " ".join(out.split())


def prepare_data(data, num_labels, testing=False, nsamples=-1, seed=None):
    """reference models/utils.py:133-200: --testing samples 200 rows; split column -> train/val/test; one-hot labels;
    sklearn-'balanced' class weights from the training split.  Like the reference, the subsamples are drawn from numpy's
    GLOBAL stream (run_mm_late.py:49 seeds it), so the ITM negative sampling that follows continues from the same state as
    in the reference (pinned by tests/golden/f3_prepare_data.json); `seed` = an explicit RandomState instead."""
    rs = None if seed is None else np.random.RandomState(seed)
    if testing:
        data = data.sample(min(200, len(data)), random_state=rs)
    parts = {s: data[data.split == s] for s in ("train", "val", "test")}
    if nsamples > 0:
        parts["train"] = parts["train"].sample(min(nsamples, len(parts["train"])), random_state=rs)
    onehot = lambda df: np.eye(num_labels, dtype=np.int64)[df.label.values.astype(int)]
    weights = balanced_class_weights(parts["train"].label.values, num_labels)
    return parts["train"], onehot(parts["train"]), parts["val"], onehot(parts["val"]), parts["test"], onehot(parts["test"]), weights


class MM_Dataset(torch.utils.data.Dataset):
    """item layout of reference models/datasets.py:125-190"""

    def __init__(self, tweet_ids, texts, labels, tokenizer, max_length, img_file_fmt, image=224, raw_images=False, batch_tokenize=False):
        self.ids, self.texts, self.labels = tweet_ids, texts, labels
        self.tok, self.max_length, self.fmt, self.image = tokenizer, max_length, img_file_fmt, image
        self.raw_images = raw_images        # True: yield the decoded RGB bytes; resize + normalize run on the GPU per batch
        self.batch_tokenize = batch_tokenize    # True: yield the normalised text; BatchTokenizeCollate encodes the whole batch in one call
        self._lut = None

    def __len__(self):
        return len(self.ids)

    def _open(self, tid):
        from PIL import Image
        path = self.fmt.format(tid)
        if not os.path.exists(path):
            path = path.replace(".jpg", ".png")               # reference :164-167 jpg -> png fallback
        return Image.open(path).convert("RGB")

    def _pixels(self, tid):
        """host form of the ViT feature extractor defaults (PIL BILINEAR resize, float64 rescale -> float32, normalize)"""
        from PIL import Image
        img = self._open(tid).resize((self.image, self.image), Image.BILINEAR)
        if self._lut is None:
            x = (np.arange(256).astype(np.float64) * (1 / 255)).astype(np.float32)
            self._lut = (x - np.float32(0.5)) / np.float32(0.5)
        return torch.from_numpy(self._lut[np.asarray(img)]).permute(2, 0, 1).contiguous().unsqueeze(0)

    def __getitem__(self, i):
        item = {"labels": torch.from_numpy(self.labels[i]), "data_id": torch.tensor(int(self.ids[i]))}
        if self.batch_tokenize:
            item["text"] = normalize_tweet(self.texts[i])
        else:
            enc = self.tok(normalize_tweet(self.texts[i]), padding="max_length", truncation=True, max_length=self.max_length, return_tensors="pt")
            item["input_ids"], item["attention_mask"] = enc["input_ids"], enc["attention_mask"]
        if self.raw_images:
            item["image"] = np.asarray(self._open(self.ids[i]))
        else:
            item["pixel_values"] = self._pixels(self.ids[i])
        return item


class Lxmert_Dataset(torch.utils.data.Dataset):
    """Item layout and file layout of the reference's Lxmert_Dataset (models/datasets.py:255-301): the tweet text tokenised per item
    ([1, T] ids / attention mask / token types) and the post's 36 pre-extracted region features and normalised boxes, one `torch.save`d
    tensor per post and kind:
        <data_path><task_name>_img_feats/features/feat_<data_id>     float [1, 36, 2048] (or [36, 2048])
        <data_path><task_name>_img_feats/boxes/nbox_<data_id>        float [1, 36, 4]
    (the reference extracts them offline with a Faster-RCNN; here they are only read).  Task "fig" carries file names as ids (`123.jpg`):
    the numeric part is the reported data id (:259-262).  Labels are one-hot rows (prepare_data)."""

    def __init__(self, data_ids, text, labels, tokenizer, max_length, task_name, normalization=True, data_path=None):
        from .config import DATA_PATH
        self.data_ids, self.text, self.labels = data_ids, text, labels
        self.task_name = task_name
        self.data_ids_num = [float(str(x).split(".")[0]) for x in data_ids] if task_name == "fig" else data_ids
        self.tokenizer, self.max_length, self.normalization = tokenizer, max_length, normalization
        self.root = (DATA_PATH if data_path is None else data_path) + "{}_img_feats/".format(task_name)

    def __len__(self):
        return len(self.labels)

    def _load(self, kind, stem, data_id):
        path = self.root + "{}/{}_{}".format(kind, stem, data_id)
        if not os.path.exists(path):
            raise FileNotFoundError(f"ROI-feature file {path!r} not found (layout of models/datasets.py:291-294)")
        try:
            t = torch.load(path, map_location="cpu", weights_only=True)
        except TypeError:                                     # older torch: no weights_only
            t = torch.load(path, map_location="cpu")
        return t.to(torch.float32)

    def __getitem__(self, index):
        text = normalize_tweet(self.text[index]) if self.normalization else self.text[index]
        enc = self.tokenizer(text, padding="max_length", max_length=self.max_length, truncation=True, return_token_type_ids=True,
                             return_attention_mask=True, add_special_tokens=True, return_tensors="pt")
        data_id = self.data_ids[index]
        feats, boxes = self._load("features", "feat", data_id), self._load("boxes", "nbox", data_id)
        if feats.shape[-2:] != (36, 2048) or boxes.shape[-2:] != (36, 4):
            raise ValueError(f"post {data_id}: features {tuple(feats.shape)} / boxes {tuple(boxes.shape)}, expected [.., 36, 2048] / [.., 36, 4]")
        return {"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"], "token_type_ids": enc["token_type_ids"],
                "features": feats.reshape(36, 2048), "normalized_boxes": boxes.reshape(36, 4),
                "labels": torch.as_tensor(self.labels[index], dtype=torch.long), "data_id": torch.tensor(int(self.data_ids_num[index]), dtype=torch.long)}


def lxmert_loaders_from_data_key(cfg, args, tokenizer, data_path=None):
    """reference MMEarly_Model.load_data, LXMERT branch (models/mm_early.py:228-258): prepare_data -> three Lxmert_Datasets -> loaders
    (train shuffled).  Data parallel: the training set is sharded by a DistributedSampler, validation / test stay whole."""
    multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    tr, ytr, va, yva, te, yte, w = prepare_data(cfg.data, cfg.num_labels, args.testing, -1, args.seed if multi else None)
    mk = lambda df, y: Lxmert_Dataset(df.tweet_id.values, df.text.values, y, tokenizer, cfg.max_length, cfg.task_name, data_path=data_path)
    kw = dict(num_workers=int(getattr(args, "num_workers", 0) or 0))
    if kw["num_workers"] > 0:
        kw.update(persistent_workers=True, prefetch_factor=4, worker_init_fn=worker_init)
    dl = lambda ds, sh: torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=sh, **kw)
    train_ds = mk(tr, ytr)
    if multi:
        sampler = torch.utils.data.distributed.DistributedSampler(train_ds, shuffle=True, seed=int(args.seed), drop_last=False)
        train_loader = torch.utils.data.DataLoader(train_ds, batch_size=cfg.batch_size, sampler=sampler, **kw)
    else:
        train_loader = dl(train_ds, True)
    return train_loader, dl(mk(va, yva), False), dl(mk(te, yte), False), w


def worker_init(_worker_id):
    """DataLoader worker: one thread per worker.  The GPU box grants a CPU share (16 cores), not the host's 256: a Rust tokenizer pool or an
    OpenMP team sized by the host's core count inside each of 8+ workers oversubscribes that share and throttles the process that enqueues
    the GPU step (tools/loader_bench.py)."""
    os.environ["TOKENIZERS_PARALLELISM"] = "false"
    os.environ["OMP_NUM_THREADS"] = "1"
    torch.set_num_threads(1)


class BatchTokenizeCollate:
    """collate_fn: the batch's normalised texts -> input_ids / attention_mask [B, max_length] in ONE tokenizer call (the fast tokenizers'
    encode_batch: Rust, all cores of the worker) instead of one Python-level call per item (reference models/datasets.py:141-146 encodes per
    item and the default collate stacks [B, 1, T]; same ids, same padding='max_length' / truncation).  The remaining fields go to `inner`
    (RawImageCollate for decoded images) or to the default collate."""

    def __init__(self, tokenizer, max_length, inner=None):
        self.tok, self.max_length, self.inner = tokenizer, max_length, inner

    def __call__(self, items):
        from torch.utils.data import default_collate
        enc = self.tok([it["text"] for it in items], padding="max_length", truncation=True, max_length=self.max_length, return_tensors="pt")
        rest = [{k: v for k, v in it.items() if k != "text"} for it in items]
        batch = self.inner(rest) if self.inner is not None else default_collate(rest)
        batch["input_ids"], batch["attention_mask"] = enc["input_ids"], enc["attention_mask"]
        return batch


def loaders_from_data_key(cfg, args, trainer):
    from transformers import AutoTokenizer
    tdir = MODEL_DIR_DICT[args.txt_model_name]
    if not os.path.isdir(tdir):
        raise FileNotFoundError(f"tokenizer directory {tdir!r} (config.MODEL_DIR_DICT) not found: use --synthetic, or place the model there")
    tok = AutoTokenizer.from_pretrained(tdir)
    # single process: numpy's global stream, exactly like the reference (run_mm_late.py:49 seeded it; the ITM sampling then
    # continues from the reference's state).  Data parallel: every rank must draw the SAME subsample, and each rank's global
    # stream is seeded seed + rank for its own ITM draws -- so an explicit RandomState(seed), which picks the same rows.
    multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
    tr, ytr, va, yva, te, yte, w = prepare_data(cfg.data, cfg.num_labels, args.testing, args.nsamples, args.seed if multi else None)
    # image resize + normalize run on the GPU per batch (image_processing.py) unless --cpu_preprocess asks for the host form
    gpu = not getattr(args, "cpu_preprocess", False) and trainer.device.type == "cuda"
    size = trainer.model.arch["image"]
    # texts are encoded per batch in the collate (one encode_batch call) unless --item_tokenize keeps the reference's per-item calls
    batch_tok = not getattr(args, "item_tokenize", False) and getattr(tok, "is_fast", False)
    mk = lambda df, y: MM_Dataset(df.tweet_id.values, df.text.values, y, tok, cfg.max_length, cfg.img_fmt, size, raw_images=gpu, batch_tokenize=batch_tok)
    kw = dict(num_workers=int(getattr(args, "num_workers", 0) or 0))
    # (Round 4 could cut a batch into sub-batches decoded by several workers side by side, MMHIP_LOADER_SPLIT: it shortened an epoch's first batch
    # from 64 to 36 ms and cost 3 % of the steady rate -- a loss over any epoch of more than a few hundred posts; removed in round 5.  What it was
    # after is done by starting the next epoch's first batches early: MMLate_Model.train, MMHIP_EPOCH_PREFETCH.)
    if kw["num_workers"] > 0:
        kw.update(persistent_workers=True, prefetch_factor=4, worker_init_fn=worker_init)
    # MMHIP_EPOCH_PREFETCH=1 (opt-in): a second training loader (its own workers) on which MMLate_Model.train starts epoch N + 1 while epoch N's
    # last batches are still being trained on -- the GPU no longer waits ~60 ms at every epoch start for the workers' first decoded batches
    # (DESIGN.md 8 f2).  Opt-in because the next epoch's shuffle is then drawn from torch's global generator BEFORE the validation / test loaders
    # draw their seeds instead of after: a seeded run visits the posts in another order than without the flag (and than the reference).
    twin_wanted = kw["num_workers"] > 0 and os.environ.get("MMHIP_EPOCH_PREFETCH", "0") == "1"
    inner, ring_collates = None, None
    if gpu:
        from .image_processing import GpuImageProcessor, RawImageCollate, RingCollate, SharedImageRing
        trainer.image_processor = GpuImageProcessor(size=size, device=trainer.device)
        inner = RawImageCollate(trainer.image_processor)
        if kw["num_workers"] > 0 and os.environ.get("MMHIP_IMAGE_RING", "1") != "0":
            # decoded images reach the training process through a pinned shared-memory ring instead of the DataLoader's result queue
            # (image_processing.SharedImageRing); a batch that does not fit its slot falls back to the queue
            slot_mb = float(os.environ.get("MMHIP_RING_SLOT_MB", str(1.5 * cfg.batch_size)))
            n_live = 2 if twin_wanted else 1                                                     # loaders that can have batches outstanding at once
            slots = min(kw["num_workers"] * 4 * n_live + 6, max(4, int(4096 / slot_mb)))         # at most 4 GB of pinned shared memory
            # the ring lives in /dev/shm: pages that do not fit there are a SIGBUS when a worker first touches them (a container with a small
            # /dev/shm) -- shrink the ring to 80 % of what is free, or go without it
            try:
                vfs = os.statvfs("/dev/shm")
                room_mb = vfs.f_bavail * vfs.f_frsize / (1 << 20) * 0.8
            except OSError:
                room_mb = float("inf")
            if slots * slot_mb > room_mb:
                fit = int(room_mb / slot_mb)
                logger.warning("image ring: %d slots of %.0f MB do not fit /dev/shm (%.0f MB usable): %s", slots, slot_mb, room_mb,
                               f"ring cut to {fit} slots" if fit >= 3 + kw["num_workers"] * n_live else "decoded images go through the DataLoader queue instead")
                slots = fit
            # every batch the loaders can have outstanding needs a slot, beside the 3 the prefetcher holds (SharedImageRing: why)
            pf = min(4, (slots - 3) // (kw["num_workers"] * n_live))
            if pf >= 1:
                kw["prefetch_factor"] = pf
                trainer.image_ring = SharedImageRing(slots, int(slot_mb * (1 << 20)))
                ring_collates = [RingCollate(trainer.image_processor, trainer.image_ring, owner=o) for o in range(4)]      # train / val / test / train's twin: one lease owner each
        kw["collate_fn"] = inner
    if batch_tok:
        kw["collate_fn"] = BatchTokenizeCollate(tok, cfg.max_length, inner)
    n_made = [0]

    def dl(ds, sh, sampler=None, owner=None):
        if ring_collates is not None:
            rc = ring_collates[min(n_made[0], 2) if owner is None else owner]
            n_made[0] += owner is None
            kw["collate_fn"] = BatchTokenizeCollate(tok, cfg.max_length, rc) if batch_tok else rc
        return torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=sh if sampler is None else None, sampler=sampler, **kw)
    train_ds = mk(tr, ytr)
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        # data parallel: every rank trains on its own shard (equal length on all ranks, so the per-stage all-reduces pair up);
        # MMLate_Model.train calls sampler.set_epoch.  Validation / test stay whole on every rank (rank 0 writes the files).
        mk_sampler = lambda: torch.utils.data.distributed.DistributedSampler(train_ds, shuffle=True, seed=int(args.seed), drop_last=False)
        train_loader = dl(train_ds, None, mk_sampler())
        if twin_wanted:
            train_loader.mmhip_twin = dl(train_ds, None, mk_sampler(), owner=3)
    else:
        train_loader = dl(train_ds, True)
        if twin_wanted:
            train_loader.mmhip_twin = dl(train_ds, True, owner=3)
    return train_loader, dl(mk(va, yva), False), dl(mk(te, yte), False), w
