class _Amp:
    def __getattr__(self, name):
        raise NotImplementedError("shim: apex.amp is only used with --use-amp (off by default)")


amp = _Amp()
