import sys, runpy, torch
torch.backends.cudnn.deterministic = True
sys.argv = ["bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"]
runpy.run_path(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "bench.py"), run_name="__main__")
