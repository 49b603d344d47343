python - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "custom-diffusion360_amd")]
import torch
from cd360 import _host, ops
h = _host.get()
print("host glue:", h is not None, "stream equal:", h.current_stream() == ops._stream())
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    print("side stream equal:", h.current_stream() == ops._stream())
PY
python -m pytest tests/test_linear_gpu.py tests/test_backward_gpu.py tests/test_capture_gpu.py -q 2>&1 | grep -a "passed\|failed"
for i in 1 2; do python tools/probe/train_ab.py 2>&1 | tail -1; CD360_NO_HOST_GLUE=1 python tools/probe/train_ab.py 2>&1 | tail -1; done
