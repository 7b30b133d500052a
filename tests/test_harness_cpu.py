"""CPU: the callers around the hot path (picklebot_amd/harness.py) against a direct restatement of the reference's
semantics: dataloader.py:10-23 (custom_collate pads T with zeros to the longest clip) and train.py:110-153
(calculate_accuracy / estimate_loss: mean loss over batches, accuracy over samples)."""
import torch
import torch.nn.functional as F

from picklebot_amd import harness


def _ref_collate(batch):
    """dataloader.py:10-23, restated."""
    videos, labels = zip(*batch)
    max_length = max(v.shape[0] for v in videos)
    padded = [F.pad(v, (0, 0, 0, 0, 0, 0, 0, max_length - v.shape[0]), value=0) for v in videos]
    return torch.stack(padded), torch.tensor(labels, dtype=torch.long)


def test_custom_collate_pads_time_with_zeros():
    g = torch.Generator().manual_seed(0)
    batch = [(torch.randint(0, 256, (t, 6, 5, 3), dtype=torch.uint8, generator=g), lab) for t, lab in ((7, 1), (4, 0), (9, 2))]
    got, labels = harness.custom_collate(batch)
    want, wl = _ref_collate(batch)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (3, 9, 6, 5, 3)
    assert torch.equal(got, want) and torch.equal(labels, wl) and labels.dtype == torch.long
    assert int(got[1, 4:].sum()) == 0                     # the padded tail of the short clip


def test_extract_features_labels_layout():
    clips = torch.randint(0, 256, (2, 4, 6, 5, 3), dtype=torch.uint8)
    feats, labels = harness.extract_features_labels((clips, torch.tensor([1, 0])), "cpu", keep_uint8=False)
    want = clips.permute(0, -1, 1, 2, 3).to(torch.bfloat16) / 255          # train.py:106
    assert torch.equal(feats, want) and feats.stride() == want.stride() and tuple(feats.shape) == (2, 3, 4, 6, 5)
    raw, _ = harness.extract_features_labels((clips, torch.tensor([1, 0])), "cpu")
    assert raw.dtype == torch.uint8 and torch.equal(raw.float() / 255, clips.permute(0, -1, 1, 2, 3).float() / 255)


class _Tiny(torch.nn.Module):
    def __init__(self, ncls):
        super().__init__()
        self.fc = torch.nn.Linear(3, ncls)

    def forward(self, x):                                  # (B,3,T,H,W) any dtype -> logits
        return self.fc(x.float().mean((2, 3, 4)))


def test_estimate_loss_matches_reference_loop():
    torch.manual_seed(1)
    model = _Tiny(4).train()
    loader = [(torch.randint(0, 256, (b, 3, 4, 4, 3), dtype=torch.uint8), torch.randint(0, 4, (b,))) for b in (5, 3, 6)]
    crit = torch.nn.CrossEntropyLoss()
    # train.py:123-153 restated: loss summed per batch then / len(loader); correct / samples
    val_loss, correct, samples = 0.0, 0, 0
    with torch.no_grad():
        for clips, labels in loader:
            out = model(clips.permute(0, -1, 1, 2, 3))
            correct += int((out.argmax(1) == labels).sum())
            samples += labels.size(0)
            val_loss += crit(out, labels).item()
    loss, acc = harness.estimate_loss(model, loader, crit, "cpu", use_autocast=False)
    assert abs(loss - val_loss / len(loader)) < 1e-6 and abs(acc - correct / samples) < 1e-12
    assert model.training                                  # mode restored
    # BCE flavour (train.py:116-121)
    bmodel = _Tiny(1)
    bl = [(c, (l > 1).long()) for c, l in loader]
    bcrit = torch.nn.BCEWithLogitsLoss()
    loss_b, acc_b = harness.estimate_loss(bmodel, bl, bcrit, "cpu", use_autocast=False)
    tot, cor, n = 0.0, 0, 0
    with torch.no_grad():
        for clips, labels in bl:
            out = bmodel(clips.permute(0, -1, 1, 2, 3))
            y = labels.unsqueeze(1).float()
            tot += bcrit(out, y).item()
            cor += int(((torch.sigmoid(out) >= 0.5).float() == y).sum())
            n += labels.size(0)
    assert abs(loss_b - tot / len(bl)) < 1e-6 and abs(acc_b - cor / n) < 1e-12


def test_prefetcher_passes_batches_through_on_cpu():
    loader = [(torch.full((2, 3), i, dtype=torch.uint8), torch.tensor([i, i])) for i in range(4)]
    got = list(harness.DevicePrefetcher(loader, "cpu"))
    assert len(got) == 4 and all(int(c[0, 0]) == i and int(l[0]) == i for i, (c, l) in enumerate(got))


def test_strip_compile_prefix():
    sd = {"_orig_mod.block1.0.weight": 1, "head.bias": 2}
    assert harness.strip_compile_prefix(sd) == {"block1.0.weight": 1, "head.bias": 2}
