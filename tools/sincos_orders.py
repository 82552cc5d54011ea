"""Counts, over every binary32 angle in [0, 6.2832], the inputs for which the two evaluation orders of the binary64
sine / cosine (separate multiplies and adds; fused multiply-adds) round to different binary32 values: python3 tools/sincos_orders.py"""
import ctypes as C, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_amd as A
L=A.lib()
L.orbx_debug_sincos_diff.restype=C.c_longlong
L.orbx_debug_sincos_diff.argtypes=[C.c_uint32,C.c_ulonglong,C.c_void_p,C.c_void_p]
import struct
hi=struct.unpack('<I',struct.pack('<f',6.2832))[0]
first=(C.c_uint32*16)(); done=C.c_ulonglong(0)
n=L.orbx_debug_sincos_diff(0,hi+1,first,C.byref(done))
print("inputs",hi+1,"compared",done.value,"differences",n,[hex(x) for x in first[:min(n,16)]])
import time
t=time.time(); n=L.orbx_debug_sincos_diff(0,hi+1,first,C.byref(done)); print("second pass: %.3f s, differences %d" % (time.time()-t, n))
print("last error:", L.orbx_last_error())
