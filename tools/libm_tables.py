"""List the constant tables of glibc's flt-32 powf / logf (as used by strelka_amd/csrc/libm_flt32.h) from the C library
of this machine: __logf_data, __powf_log2_data and __exp2f_data are located in libm.so.6 by their first entries (they
are internal symbols) and printed as hex floats.  Provenance / verification aid only; nothing imports this."""
import struct
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
d = open(path, "rb").read()


def find_all(b):
    out, i = [], 0
    while True:
        i = d.find(b, i)
        if i < 0:
            return out
        out.append(i)
        i += 1


first = struct.pack("<d", float.fromhex("0x1.661ec79f8f3bep+0"))  # 1/c of the first sub-interval: logf, log2f, powf tables
hits = find_all(first)
print("tables starting with 1/c0:", hits)
for off in hits:
    v = struct.unpack_from("<40d", d, off)
    print(off, "logc0 =", float.hex(v[1]), " after the table:", [float.hex(x) for x in v[32:37]])
e = find_all(struct.pack("<QQ", 0x3FF0000000000000, 0x3FEFD9B0D3158574))
print("__exp2f_data at", e)
for off in e:
    print([hex(x) for x in struct.unpack_from("<32Q", d, off)])
    print([float.hex(x) for x in struct.unpack_from("<9d", d, off + 256)])
