"""Import-name drop-in: ``import speecht.speech_model`` etc. resolve to the MI355X-native modules of ``speecht_amd``.

The reference's callers import ``speecht.<module>`` (execution.py:20-23, evaluation.py:20-23, speecht-cli:163-205); with
this package on the path in place of the reference's, they run unmodified on the HIP path.  Modules the reference has and
this path leaves out of scope (corpus download, microphone recording, KenLM parameter search) are not
aliased: importing them raises ImportError, never a silent stand-in."""
import importlib
import sys

_MODULES = ('vocabulary', 'preprocessing', 'speech_input', 'speech_model', 'evaluation', 'training', 'execution', 'exporting')
_OUT_OF_SCOPE = ('corpus', 'recording', 'record_utils', 'parameter_search')


def __getattr__(name):
  if name in _MODULES:
    module = importlib.import_module('speecht_amd.' + name)
    sys.modules[__name__ + '.' + name] = module
    return module
  if name in _OUT_OF_SCOPE:
    raise ImportError('speecht.{} is outside the MI355X-native path (DESIGN.md section 7)'.format(name))
  raise AttributeError(name)


class _AliasFinder:
  """``import speecht.speech_model`` / ``from speecht.speech_model import X``: hand the import system the speecht_amd module."""

  @staticmethod
  def find_spec(fullname, path=None, target=None):
    head, _, tail = fullname.partition('.')
    if head != __name__ or tail not in _MODULES:
      return None
    import importlib.util

    class _Loader:
      real_spec = None

      @classmethod
      def create_module(cls, spec):
        module = importlib.import_module('speecht_amd.' + tail)
        cls.real_spec = module.__spec__
        return module

      @classmethod
      def exec_module(cls, module):
        # the import machinery has just stamped the ALIAS spec onto the speecht_amd module it was handed: put the module's own
        # spec back, or `importlib.reload(speecht_amd.<module>)` would go looking for a loader that loads nothing
        module.__spec__ = cls.real_spec
    return importlib.util.spec_from_loader(fullname, _Loader())


sys.meta_path.insert(0, _AliasFinder)
