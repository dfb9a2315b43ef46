"""GNU Radio base classes for the drop-in blocks.

With GNU Radio installed the blocks derive from the real ``gnuradio.gr`` classes (the scheduler calls
``work(input_items, output_items)`` / ``general_work`` with NumPy views of its stream buffers).  Without it (this build
container, the GPU test box) stand-ins with the same constructors keep the classes importable so they
can be driven the way the reference's own offline harness does it
(`TEST/GNU_RADIO_OFFLINE/topblock.py:84-88`: call ``work()`` by hand).
"""
try:  # pragma: no cover - GNU Radio is not installed in the build/test images
    from gnuradio import gr as _gr

    sync_block = _gr.sync_block
    basic_block = _gr.basic_block
    decim_block = _gr.decim_block
    HAVE_GNURADIO = True
except Exception:  # ImportError or a broken install
    HAVE_GNURADIO = False

    class _block:
        def __init__(self, name=None, in_sig=None, out_sig=None):
            self._name = name
            self._in_sig = in_sig
            self._out_sig = out_sig
            self._output_multiple = 1
            self._relative_rate = 1.0
            self.consumed = []          # items consumed per input port by the last general_work (what the scheduler would see)

        def name(self):
            return self._name

        def set_output_multiple(self, n):
            self._output_multiple = int(n)

        def output_multiple(self):
            return self._output_multiple

        def set_relative_rate(self, r):
            self._relative_rate = float(r)

        def relative_rate(self):
            return self._relative_rate

    class sync_block(_block):  # noqa: N801  (GNU Radio's name)
        pass

    class decim_block(_block):  # noqa: N801
        def __init__(self, name=None, in_sig=None, out_sig=None, decim=1):
            _block.__init__(self, name, in_sig, out_sig)
            self._decim = int(decim)
            self._relative_rate = 1.0 / self._decim

        def decimation(self):
            return self._decim

    class basic_block(_block):  # noqa: N801
        def consume(self, port, n):
            while len(self.consumed) <= port:
                self.consumed.append(0)
            self.consumed[port] = int(n)

        def consume_each(self, n):
            self.consumed = [int(n)] * max(1, len(self._in_sig or []))
