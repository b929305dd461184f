import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genomeworks_amd import cudaaligner, synthetic
for n, L in ((2000, 1000), (20000, 150)):
    pairs = synthetic.generate_pairs(1, n, L, L // 30, L // 30, L // 30)
    al = cudaaligner.CudaAlignerBatch(L + L // 10 + 8, L + L // 10 + 8, n, max_device_memory_allocator_caching_size=16 << 30)
    for q, t in pairs:
        assert al.add_alignment(q, t) == 0
    t0 = time.perf_counter(); al.align_all(); al.sync(); t1 = time.perf_counter()
    al2 = None
    print("default aligner: %d pairs x %d bp: %.1f ms (align_all + sync) -> %.0f pairs/s" % (n, L, (t1 - t0) * 1e3, n / (t1 - t0)))
