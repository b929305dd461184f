"""Index-split sharding of independent work units (POA windows / alignment pairs) over the GPUs of one node.

Windows and pairs never communicate, so there is no data-path collective: every rank (one process per GPU,
`torch.distributed`) takes a contiguous slice of the unit indices, runs it through its own Batch / Aligner, and
results are placed by GLOBAL index, so the output does not depend on the number of ranks. The only collective is
the optional gather of result objects onto rank 0 (host side; RCCL is not involved in the data path).
Reference context: the reference never splits one batch over devices; its tools run one worker per device pulling
whole batches (cudamapper/src/main.cu:577-592)."""
import os


def shard_range(n_units, rank, world_size):
    """Contiguous, balanced slice [lo, hi) of range(n_units) for `rank` (first n % world ranks get one more)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(n_units, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def balanced_partition(costs, world_size):
    """Cost-balanced split: units sorted by estimated cost, heaviest first, each dealt to the
    currently lightest rank (LPT). Returns world_size ascending index lists. Deterministic (ties by index), so
    every rank computes the same partition without communicating. The reference bins windows by size for the same
    reason (get_multi_batch_sizes, cudapoa/src/utils.cu:66-146) and sorts pairs by length
    (aligner_global_myers_banded.cpp:306-309)."""
    if world_size <= 0:
        raise ValueError("bad world_size")
    import heapq
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    heap = [(0, r) for r in range(world_size)]
    parts = [[] for _ in range(world_size)]
    for i in order:
        load, r = heapq.heappop(heap)
        parts[r].append(i)
        heapq.heappush(heap, (load + costs[i], r))
    for part in parts:
        part.sort()
    return parts


def poa_window_cost(reads, band_width=256):
    """Estimated DP cells of one window: every read after the first against a graph of about the backbone length,
    band_width columns per row (0 = full band: the read length)."""
    if len(reads) < 2:
        return 0
    rows = len(reads[0])
    return sum(rows * (band_width if band_width else len(r)) for r in reads[1:])


def pair_cost(query, target):
    """Estimated cost of one alignment pair (the reference's scheduling key: query + target length)."""
    return len(query) + len(target)


def dist_info():
    """(rank, local_rank, world_size) from the torchrun environment (1 process: (0, 0, 1))."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


_HOST_GROUP = None


def host_group():
    """Process group for HOST objects (pickled result lists): gloo. With the RCCL ("nccl") default group an object
    collective would stage its pickles through device tensors; RCCL only ever carries the barrier and the timing scalars.
    None when the default group is gloo already (or no group is initialised). Collective: every rank must call it."""
    global _HOST_GROUP
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_backend() == "gloo":
        return None
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.new_group(backend="gloo")
    return _HOST_GROUP


def run_sharded(units, process_fn, gather=True, costs=None):
    """Process `units` (a list) with process_fn(list_of_units, lo) -> list of per-unit results on every rank's
    own slice. Returns the full result list in global order on rank 0 (None elsewhere) when gather is True and a
    process group is initialised; otherwise the local slice results. With `costs` (one number per unit) the split
    is cost-balanced instead of contiguous and process_fn receives (list_of_units, list_of_global_indices)."""
    import torch.distributed as dist
    active = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if active else 0
    world = dist.get_world_size() if active else 1
    if costs is not None:
        if len(costs) != len(units):
            raise ValueError("one cost per unit")
        mine = balanced_partition(costs, world)[rank]
        local = process_fn([units[i] for i in mine], mine)
        if len(local) != len(mine):
            raise RuntimeError("process_fn must return one result per unit")
        if not (active and gather):
            return local
        parts = [None] * world if rank == 0 else None
        dist.gather_object((mine, local), parts, dst=0, group=host_group())
        if rank != 0:
            return None
        out = [None] * len(units)
        for idx, res in parts:
            for i, r in zip(idx, res):
                out[i] = r
        return out
    lo, hi = shard_range(len(units), rank, world)
    local = process_fn(units[lo:hi], lo)
    if len(local) != hi - lo:
        raise RuntimeError("process_fn must return one result per unit")
    if not (active and gather):
        return local
    parts = [None] * world if rank == 0 else None
    dist.gather_object((lo, local), parts, dst=0, group=host_group())
    if rank != 0:
        return None
    out = [None] * len(units)
    for plo, res in parts:
        out[plo:plo + len(res)] = res
    return out
