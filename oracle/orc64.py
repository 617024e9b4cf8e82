"""The ctypes wrapper of oracle/orc.py over liborc_f64.so: art_oracle.c compiled with every float a double (oracle/Makefile).
TEST INFRASTRUCTURE, like the rest of oracle/.  Same API as orc (arrays are float64); no packing / presentation."""
import os

_F64 = True
_src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "orc.py")
exec(compile(open(_src).read(), _src, "exec"))
