"""The callers on either side of the hot path, host-side PyTorch:

  custom_collate              dataloader.py:10-23   variable-length clips zero-padded on T to the longest, stacked (B,T,H,W,C)
  extract_features_labels     train.py:102-108      H2D + the /255 cast - here the clip STAYS uint8 (the stem kernel folds
                                                    the /255 into its load: one pass less over the largest tensor)
  DevicePrefetcher            train.py:62-72,102    pinned-memory H2D on a copy stream, one batch ahead of compute
  calculate_accuracy(_bce)    train.py:110-121
  estimate_loss               train.py:123-153      eval loop: mean loss over batches, accuracy over samples - accumulated on
                                                    the device, ONE host sync at the end (the reference syncs twice per
                                                    batch with .item(); same numbers)

Nothing here launches a HIP kernel of its own; it is the glue a `train.py` needs around the drop-in modules.
"""
import torch
import torch.nn.functional as F


def custom_collate(batch, pin_memory=False):
    """[(video uint8 (T,H,W,C), label)] -> (padded (B,Tmax,H,W,C) uint8, labels int64): dataloader.py:10-23."""
    videos, labels = zip(*batch)
    max_length = max(v.shape[0] for v in videos)
    out = torch.zeros((len(videos), max_length) + tuple(videos[0].shape[1:]), dtype=videos[0].dtype,
                      pin_memory=pin_memory and torch.cuda.is_available())
    for i, v in enumerate(videos):
        if tuple(v.shape[1:]) != tuple(videos[0].shape[1:]):
            raise ValueError("custom_collate: clips of one batch must share (H, W, C)")
        out[i, :v.shape[0]] = v                       # the tail stays zero, like F.pad(..., value=0)
    return out, torch.tensor(labels, dtype=torch.long)


def extract_features_labels(output, device, keep_uint8=True, criterion="ce"):
    """(clips (B,T,H,W,C) uint8, labels) -> (features logical (B,C,T,H,W), labels) on `device` (train.py:102-108).
    keep_uint8: hand the raw bytes to the model (picklebot_amd's stems scale by 1/255 while loading); otherwise the
    reference's bf16 tensor.  Labels: (B,) int64 for cross-entropy, (B,1) float for BCE (train.py:107,141)."""
    clips = output[0].to(device, non_blocking=True)
    feats = clips.permute(0, -1, 1, 2, 3)              # channels_last_3d view, no copy
    if not keep_uint8:
        feats = feats.to(torch.bfloat16) / 255
    labels = output[1].to(device, non_blocking=True)
    if criterion != "ce":
        labels = labels.unsqueeze(1).float()
    return feats, labels


class DevicePrefetcher:
    """Iterate a DataLoader one batch ahead: each (clips, labels) pair is copied host -> device from pinned memory on a
    side stream while the previous batch computes; the consumer's stream waits on the copy's event only."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def _stage(self, batch):
        clips, labels = batch
        if self.stream is None:
            return clips.to(self.device), labels.to(self.device), None
        if not clips.is_pinned():
            clips = clips.pin_memory()
        with torch.cuda.stream(self.stream):
            d_clips = clips.to(self.device, non_blocking=True)
            d_labels = labels.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return d_clips, d_labels, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            clips, labels, ev = cur
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                clips.record_stream(torch.cuda.current_stream(self.device))
                labels.record_stream(torch.cuda.current_stream(self.device))
            yield clips, labels

    def __len__(self):
        return len(self.loader)


@torch.no_grad()
def calculate_accuracy(outputs, labels):
    """Number of correct top-1 predictions (a 0-dim tensor, train.py:110-114)."""
    return torch.sum(torch.argmax(outputs, dim=1) == labels)


@torch.no_grad()
def calculate_accuracy_bce(outputs, labels, threshold=0.5):
    """train.py:116-121 (kept on the device: no .cpu()/.item() per batch)."""
    return torch.sum((torch.sigmoid(outputs.float()) >= threshold).float() == labels.float())


@torch.no_grad()
def estimate_loss(model, val_loader, criterion, device, use_autocast=True, dtype=torch.bfloat16, keep_uint8=True):
    """-> (mean loss over batches, accuracy over samples) as Python floats (train.py:123-153)."""
    was_training = model.training
    model.eval()
    bce = isinstance(criterion, torch.nn.BCEWithLogitsLoss)
    dev = torch.device(device)
    correct = torch.zeros((), dtype=torch.float64, device=dev)
    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    samples, batches = 0, 0
    for output in val_loader:
        feats, labels = extract_features_labels(output, dev, keep_uint8=keep_uint8, criterion="bce" if bce else "ce")
        if use_autocast and dev.type == "cuda":
            with torch.autocast(device_type="cuda", dtype=dtype):
                outputs = model(feats)
                loss = criterion(outputs, labels)
        else:
            outputs = model(feats)
            loss = criterion(outputs, labels)
        correct += (calculate_accuracy_bce if bce else calculate_accuracy)(outputs, labels)
        loss_sum += loss.double()
        samples += labels.size(0)
        batches += 1
    model.train(was_training)
    if batches == 0:
        return float("nan"), float("nan")
    return float(loss_sum) / batches, float(correct) / samples


def strip_compile_prefix(state_dict):
    """train.py:38-44: checkpoints saved from a torch.compile'd model carry an `_orig_mod.` prefix."""
    return {k.replace("_orig_mod.", ""): v for k, v in state_dict.items()}
