"""stand-in for astropy.io (golden generation only)"""


class _Ascii:
    @staticmethod
    def write(*a, **k):
        raise RuntimeError('astropy shim: ascii.write not available')

    @staticmethod
    def read(*a, **k):
        raise RuntimeError('astropy shim: ascii.read not available')


ascii = _Ascii()
