"""Reader/writer for the next-plaid on-disk index directory, oracle side (numpy only).

Independent of the product's C++ loader (next-plaid_amd/csrc/index_loader.cpp) so the two can
check each other.  Layout follows the reference writer `write_index_from_encoded_chunks`
(index.rs:373-528) and loader `MmapIndex::load` (index.rs:1026-1139); file table in
SURVEY.md Appendix A.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import json
import os

import numpy as np


def read_index(path: str) -> dict:
    """index.rs:1026-1139 restated: chunk concatenation in chunk order, no padding rows
    (padding rows of merged_*.npy are never addressed by doc_offsets)."""
    with open(os.path.join(path, "metadata.json")) as f:
        meta = json.load(f)
    nbits = int(meta["nbits"])
    num_chunks = int(meta["num_chunks"])
    centroids = np.load(os.path.join(path, "centroids.npy")).astype(np.float32, copy=False)
    bucket_weights = np.load(os.path.join(path, "bucket_weights.npy")).astype(np.float32, copy=False)
    ivf = np.load(os.path.join(path, "ivf.npy")).astype(np.int64, copy=False)
    ivf_lengths = np.load(os.path.join(path, "ivf_lengths.npy")).astype(np.int32)  # fast-plaid writes i64
    doclens, codes, residuals = [], [], []
    for i in range(num_chunks):
        with open(os.path.join(path, f"doclens.{i}.json")) as f:
            doclens.extend(json.load(f))
        codes.append(np.load(os.path.join(path, f"{i}.codes.npy")).astype(np.int64, copy=False))
        residuals.append(np.load(os.path.join(path, f"{i}.residuals.npy")).view(np.uint8))
    pd = centroids.shape[1] * nbits // 8
    return dict(
        nbits=nbits, metadata=meta, centroids=centroids, bucket_weights=bucket_weights, ivf=ivf,
        ivf_lengths=ivf_lengths, doc_lengths=np.asarray(doclens, np.int64),
        codes=np.concatenate(codes) if codes else np.zeros(0, np.int64),
        residuals=np.concatenate(residuals, 0) if residuals else np.zeros((0, pd), np.uint8),
    )


def build_ivf(codes: np.ndarray, doc_lengths: np.ndarray, K: int):
    """index.rs:479-499: per centroid, the sorted unique doc ids holding a token with that code."""
    doc_of_tok = np.repeat(np.arange(doc_lengths.size, dtype=np.int64), doc_lengths)
    key = np.unique(codes.astype(np.int64) * np.int64(max(doc_lengths.size, 1)) + doc_of_tok)
    cen = key // max(doc_lengths.size, 1)
    ivf = (key % max(doc_lengths.size, 1)).astype(np.int64)
    ivf_lengths = np.bincount(cen, minlength=K).astype(np.int32)
    return ivf, ivf_lengths
