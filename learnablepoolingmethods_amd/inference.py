"""Inference output (reference: inference.py:88-96,182): the ``VideoId,LabelConfidencePairs`` CSV of each video's top-k
classes, and a driver that runs a trained model over reader batches.  SURVEY 8f rank 5."""
from __future__ import annotations

from typing import Iterable, Iterator, Sequence

import torch

CSV_HEADER = "VideoId,LabelConfidencePairs\n"


def format_lines(video_ids: Sequence, predictions, top_k: int) -> Iterator[str]:
    """inference.py:88-96: one line per video, ``<id>,<class> <score> <class> <score> ...`` with the top_k classes in
    descending score order, scores printed with ``%g``.  (Equal scores keep ascending class order here; the reference's
    order among ties follows numpy.argpartition.)"""
    p = torch.as_tensor(predictions)
    k = min(int(top_k), p.shape[1])
    scores, classes = torch.sort(p, dim=1, descending=True, stable=True)
    scores, classes = scores[:, :k].cpu().tolist(), classes[:, :k].cpu().tolist()
    for vid, cs, ss in zip(video_ids, classes, scores):
        if isinstance(vid, bytes):
            vid = vid.decode("utf-8")
        yield vid + "," + " ".join("%i %g" % (c, s) for c, s in zip(cs, ss)) + "\n"


def write_predictions(out_file, trainer, batches: Iterable, top_k: int = 20) -> int:
    """Header + format_lines for every (ids, frames, labels, num_frames) batch of ``readers.YT8MFrameFeatureReader.batches``;
    ``trainer.predict`` is the eval-mode forward (moving batch-norm statistics).  Returns the number of videos written."""
    out_file.write(CSV_HEADER)
    n = 0
    for ids, frames, _, num_frames in batches:
        pred = trainer.predict(frames, num_frames)
        for line in format_lines(ids, pred, top_k):
            out_file.write(line)
        n += len(ids)
    return n
