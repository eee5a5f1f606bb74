"""Multi-GPU driver logic for the sharded training set (one process per GPU).

The streams of one logical training set are dealt to ranks in contiguous blocks;
every rank keeps a full replica of the weights and optimiser state, computes the
weight deltas of its own streams, the deltas are summed over ranks with ONE
all-reduce per generation (RCCL over xGMI on the GPUs; gloo in the CPU tests),
and every rank applies the identical update.  The reference has no counterpart
(it sums the streams' deltas in one process: recur-nn.c:724-739); this is the
same sum, distributed.
"""


def shard_range(rank, world, streams_per_rank):
    """(first global stream, streams on this rank, global stream count)."""
    return rank * streams_per_rank, streams_per_rank, world * streams_per_rank


class ShardedStep:
    """One generation = local deltas -> all-reduce(sum) -> replicated update.

    compute_deltas(i): fills the local delta buffer for text position i
    all_reduce():      sums the delta buffer over ranks, in place (None: single rank)
    apply():           the optimiser step from the (now global) deltas
    """

    def __init__(self, compute_deltas, all_reduce, apply):
        self.compute_deltas = compute_deltas
        self.all_reduce = all_reduce
        self.apply = apply

    def __call__(self, i):
        self.compute_deltas(i)
        if self.all_reduce is not None:
            self.all_reduce()
        self.apply()
