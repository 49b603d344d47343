import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "custom-diffusion360_amd"), os.path.join(os.getcwd(), "tools")]
import bench_train
mine = bench_train.measure(steps=3, warmup=1, batch=4, views=4, latent=64, profile=False, library=False)
lib = bench_train.measure(steps=3, warmup=1, batch=4, views=4, latent=64, profile=False, library=True)
print("python-node" if os.environ.get("CD360_NO_HOST_GLUE") else "c++-node", "eager cd360_ms", mine["ms_per_step"], "library_ms", lib["ms_per_step"], "losses", mine["losses"][-2:])
