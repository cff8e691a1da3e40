"""A/B helper: bench.py with the grouped expand1x1 weight-gradient launch switched off (every layer its own launch)."""
import runpy, sys
sys.path.insert(0, '.')
import squeezedet_pytorch_amd  # noqa: F401
from squeezedet_pytorch_amd import ops
ops.wgrad1x1_groups = lambda *a, **k: {}
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')
