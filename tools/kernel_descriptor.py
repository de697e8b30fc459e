"""VGPR allocation the kernel descriptors of an object's gfx950 code object REQUEST (granulated count, which the backend pads
to enforce `amdgpu_waves_per_eu`'s upper bound) next to what the code uses: python tools/kernel_descriptor.py <obj.o> <substr>"""
import re, struct, subprocess, sys
B = "/opt/rocm/lib/llvm/bin/"
obj, pat = sys.argv[1], sys.argv[2]
subprocess.run([B + "llvm-objcopy", "--dump-section", ".hip_fatbin=/tmp/fb.bin", obj, "/tmp/fb_dummy.o"], check=True)
subprocess.run([B + "clang-offload-bundler", "--unbundle", "--type=o", "--input=/tmp/fb.bin",
                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=/tmp/k.co"], check=True)
sec = subprocess.run([B + "llvm-readelf", "-S", "/tmp/k.co"], capture_output=True, text=True).stdout
m = re.search(r"\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
addr, off = int(m.group(1), 16), int(m.group(2), 16)
data = open("/tmp/k.co", "rb").read()
syms = subprocess.run([B + "llvm-readelf", "-s", "-W", "/tmp/k.co"], capture_output=True, text=True).stdout
for line in syms.splitlines():
    f = line.split()
    if len(f) >= 8 and f[-1].endswith(".kd"):
        name = subprocess.run(["c++filt", f[-1][:-3]], capture_output=True, text=True).stdout.strip()
        if pat not in name:
            continue
        v = int(f[1], 16)
        kd = data[off + v - addr: off + v - addr + 64]
        rsrc1 = struct.unpack_from("<I", kd, 48)[0]
        rsrc3 = struct.unpack_from("<I", kd, 44)[0]
        print("%-80s granulated vgpr alloc %3d (waves/SIMD <= %d), accum_offset %d" % (name[:80], ((rsrc1 & 63) + 1) * 8, min(8, 512 // (((rsrc1 & 63) + 1) * 8)), ((rsrc3 & 63) + 1) * 4))
