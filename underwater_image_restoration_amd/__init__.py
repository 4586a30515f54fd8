"""MI355X-native restoration forward pass (student NAFNetWFIGate, IR-SDE steps) behind the reference's Python API."""
__all__ = ["NAFNetWFIGate"]


def __getattr__(name):
    if name == "NAFNetWFIGate":
        from .student import NAFNetWFIGate
        return NAFNetWFIGate
    raise AttributeError(name)
