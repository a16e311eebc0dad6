"""TEST INFRASTRUCTURE: builds and loads the SIMT-emulator build of the kernel sources
(tests/emu/libmapnet_emu.so).  Used by CPU tests to execute the real kernel code; never by the
product."""
import ctypes
import os
import subprocess

from geomapnet_amd._binding import Binding

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
_cached = None


def load():
    global _cached
    if _cached is None:
        subprocess.check_call(["make", "-s", "-j8", "-C", _DIR, "libmapnet_emu.so"])
        _cached = Binding(ctypes.CDLL(os.path.join(_DIR, "libmapnet_emu.so")))
        assert _cached.backend_name == "emu"
    return _cached
