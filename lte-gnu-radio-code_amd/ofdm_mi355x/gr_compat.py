"""GNU Radio base class for the drop-in blocks.

With GNU Radio installed the blocks derive from the real ``gnuradio.gr.sync_block`` (the scheduler calls
``work(input_items, output_items)`` with NumPy views of its stream buffers).  Without it (this build
container, the GPU test box) a stand-in with the same constructor keeps the classes importable so they
can be driven the way the reference's own offline harness does it
(`TEST/GNU_RADIO_OFFLINE/topblock.py:84-88`: call ``work()`` by hand).
"""
try:  # pragma: no cover - GNU Radio is not installed in the build/test images
    from gnuradio import gr as _gr

    sync_block = _gr.sync_block
    HAVE_GNURADIO = True
except Exception:  # ImportError or a broken install
    HAVE_GNURADIO = False

    class sync_block:  # noqa: N801  (GNU Radio's name)
        def __init__(self, name=None, in_sig=None, out_sig=None):
            self._name = name
            self._in_sig = in_sig
            self._out_sig = out_sig

        def name(self):
            return self._name
