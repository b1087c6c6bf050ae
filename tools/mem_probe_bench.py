"""bench.py with a report of the GPU memory it reserved (stderr, at exit): usage as bench.py, e.g. mem_probe_bench.py --pipeline 8 --no-cpu-baseline"""
import runpy, sys, torch, atexit
def report():
    try:
        free, total = torch.cuda.mem_get_info()
        print("MEM max_reserved GB %.1f max_alloc GB %.1f free now GB %.1f of %.1f" % (torch.cuda.max_memory_reserved()/2**30, torch.cuda.max_memory_allocated()/2**30, free/2**30, total/2**30), file=sys.stderr, flush=True)
    except Exception as e:
        print("MEM probe failed", e, file=sys.stderr)
atexit.register(report)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
