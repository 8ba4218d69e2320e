def parallel_sort(*a, **k):
    raise NotImplementedError("dead path in the reference (GPU_MODE is a constant False)")
