"""Synthetic posts of the benchmark shape (BASELINE.md 3 / SURVEY.md 8d): 128-token texts + 224x224 images.
ids uniform in [3, V) with CLS first / EOS last, no padding unless asked; pixels uniform(-1, 1) (the range the ViT image
processor produces: rescale 1/255, mean = std = 0.5); one-hot int64 labels."""
import torch


def synthetic_batch(vocab, num_labels, B, T=128, seed=1234, txt_kind="xlmr", pad_id=1, pad=False, image=224, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab, (B, T), generator=g, dtype=torch.int64)
    cls_id, eos_id = (0, 2) if txt_kind == "xlmr" else (101, 102)
    ids[:, 0] = cls_id
    mask = torch.ones(B, T, dtype=torch.int64)
    if pad:
        lens = torch.randint(4, T + 1, (B,), generator=g)
        lens[0] = T
        for b in range(B):
            n = int(lens[b])
            ids[b, n - 1] = eos_id
            ids[b, n:] = pad_id
            mask[b, n:] = 0
    else:
        ids[:, T - 1] = eos_id
    pixels = torch.rand(B, 3, image, image, generator=g) * 2 - 1
    labels = torch.randint(0, num_labels, (B,), generator=g)
    onehot = torch.nn.functional.one_hot(labels, num_labels).to(torch.int64)
    return ids.to(device), mask.to(device), pixels.to(device), onehot.to(device)


class SyntheticPosts(torch.utils.data.Dataset):
    """dataset with the reference MM_Dataset item layout (models/datasets.py:125-190): input_ids / attention_mask [1,T],
    pixel_values [1,3,H,W], labels one-hot int64, data_id"""

    def __init__(self, n, vocab, num_labels, T=128, seed=0, txt_kind="xlmr", pad_id=1, image=224):
        self.n, self.args = n, (vocab, num_labels, T, seed, txt_kind, pad_id, image)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        vocab, C, T, seed, kind, pad_id, image = self.args
        ids, mask, px, onehot = synthetic_batch(vocab, C, 1, T, seed * 1000003 + i, kind, pad_id, True, image)
        return {"input_ids": ids, "attention_mask": mask, "pixel_values": px, "labels": onehot[0], "data_id": torch.tensor(i)}
