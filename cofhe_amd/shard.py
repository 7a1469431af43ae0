"""Row-wise sharding of ciphertext tensors across the GPUs of one node (one process per GPU).

add_ciphertext_tensors is element-wise and row i of scal_ciphertext_tensors' 2-D result only
needs row i of the ciphertext matrix (reference loops:
include/x86_64/cpu_cryptosystem_tensor_ops.inl:242-264 and :396-417), so rank r owns a
contiguous block of rows, no exchange happens between chained operations, and the result is
reassembled with ONE all-gather of fixed-size records: cofhe_hip_all_gather_rows (csrc/shard.hip, RCCL over xGMI),
which executes the plan of cofhe_hip_gather_plan.  This module only holds the partition arithmetic bench.py and the
tests share; tests/test_shard_gloo.py executes the library's plan over gloo.
"""
from typing import List, Tuple

REC_WORDS = 168
CT_WORDS = 2 * REC_WORDS


def row_partition(n_rows: int, world: int) -> List[Tuple[int, int]]:
    """[start, stop) row ranges, remainder rows to the low ranks."""
    base, rem = divmod(n_rows, world)
    out, start = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((start, start + cnt))
        start += cnt
    return out


def rows_for_mode(rows: int, world: int, rank: int, scaling: str) -> Tuple[int, int, int]:
    """(first global row, rows on this rank, rows of the assembled tensor).  weak: every rank holds its own `rows`-row
    tensor and the assembled tensor has rows * world rows; strong: ONE `rows`-row tensor is split over the ranks."""
    if scaling == "weak":
        return rank * rows, rows, rows * world
    if scaling != "strong":
        raise ValueError("scaling must be weak or strong")
    a, b = row_partition(rows, world)[rank]
    return a, b - a, rows


def shard_records(records, n_rows: int, n_cols: int, world: int, rank: int):
    """rows [start, stop) of a row-major (n_rows x n_cols) ciphertext-record tensor (flat int32)."""
    start, stop = row_partition(n_rows, world)[rank]
    w = n_cols * CT_WORDS
    return records[start * w: stop * w]
