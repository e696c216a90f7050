"""`speecht-cli export --weights DIR`: dump the trainable variables as .npy in the reference's
directory layout (mirror of speecht/exporting.py:26-44)."""
from .speech_input import SingleInputLoader
from .speech_model import Session, create_default_model


class Exporting:

  def __init__(self, flags):
    self.flags = flags

  def run(self):
    with Session(getattr(self.flags, 'device', 'cuda:0')) as sess:
      model = create_default_model(self.flags, self.flags.input_size, SingleInputLoader(self.flags.input_size))
      model.restore(sess, self.flags.run_train_dir)
      if self.flags.export_weights_dir:
        model.export_weights(self.flags.export_weights_dir)
        return
      print('Nothing to do.')
