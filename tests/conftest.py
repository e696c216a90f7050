import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
  return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session', autouse=True)
def _tuning_overrides_from_env():
  """ST_TEST_TUNE="name=value,name=value": run the suite with library tuning overrides (st_set_tuning) -- how a kernel
  variant behind a knob is put through the same parity tests as the default before it becomes the policy."""
  spec = os.environ.get('ST_TEST_TUNE', '')
  if spec:
    from speecht_amd._lib import set_tuning
    for kv in spec.split(','):
      name, value = kv.split('=')
      set_tuning(name.strip(), int(value))
  yield
