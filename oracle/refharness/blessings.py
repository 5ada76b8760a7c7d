"""Identity stand-in for the `blessings` terminal-styling package (golden
generator only)."""


class _Style(str):
    def __call__(self, s=''):
        return s


class Terminal:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Style('')
