"""Host-side restatement of Go's math.Cos (Go standard library, src/math/sin.go —
the Cephes `cos` port) for |x| < 2**29.

The reference evaluates math.Cos(query.ConeAOI.Angle) on the host
(pkg/channeld/spatial.go:295); its value is passed to the device as
chd_aoi_query.cone_cos, so the GPU never evaluates a cosine.  A Go caller passes
Go's own math.Cos; this module is what the Python/C++ host mirror uses instead.
Python floats are IEEE binary64 and CPython never fuses a*b+c, like amd64 Go.
"""
from __future__ import annotations

import math

_PI4A = 7.85398125648498535156e-1
_PI4B = 3.77489470793079817668e-8
_PI4C = 2.69515142907905952645e-15
_SIN = (1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
        -1.98412698295895385996e-4, 8.33333333332211858878e-3, -1.66666666666666307295e-1)
_COS = (-1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
        2.48015872888517045348e-5, -1.38888888888730564116e-3, 4.16666666666665929218e-2)
_M4PI = 1.2732395447351628  # 4/Pi rounded to float64


def go_cos(x: float) -> float:
    if math.isnan(x) or math.isinf(x):
        return math.nan
    sign = False
    x = abs(x)
    if x >= float(1 << 29):
        raise ValueError("go_cos: Payne-Hanek range (|x| >= 2**29) is not restated; angles never get there")
    j = int(x * _M4PI)
    y = float(j)
    if j & 1:
        j += 1
        y += 1.0
    j &= 7
    z = ((x - y * _PI4A) - y * _PI4B) - y * _PI4C
    if j > 3:
        j -= 4
        sign = not sign
    if j > 1:
        sign = not sign
    zz = z * z
    if j == 1 or j == 2:
        y = z + z * zz * ((((((_SIN[0] * zz) + _SIN[1]) * zz + _SIN[2]) * zz + _SIN[3]) * zz + _SIN[4]) * zz + _SIN[5])
    else:
        y = 1.0 - 0.5 * zz + zz * zz * ((((((_COS[0] * zz) + _COS[1]) * zz + _COS[2]) * zz + _COS[3]) * zz + _COS[4]) * zz + _COS[5])
    return -y if sign else y
