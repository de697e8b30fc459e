import subprocess,sys,re
obj=sys.argv[1]; pat=sys.argv[2]
B="/opt/rocm/lib/llvm/bin/"
subprocess.run([B+"llvm-objcopy","--dump-section",".hip_fatbin=/tmp/fb.bin",obj,"/tmp/fb_dummy.o"],check=True)
out=subprocess.run([B+"clang-offload-bundler","--list","--type=o","--input=/tmp/fb.bin"],capture_output=True,text=True).stdout
tgt=[l for l in out.split() if "gfx950" in l][0]
subprocess.run([B+"clang-offload-bundler","--unbundle","--type=o","--input=/tmp/fb.bin","--targets="+tgt,"--output=/tmp/k.co"],check=True)
notes=subprocess.run([B+"llvm-readelf","--notes","/tmp/k.co"],capture_output=True,text=True).stdout
for blk in notes.split("- .agpr_count")[1:]:
    name=re.search(r"\.name:\s+(\S+)",blk).group(1)
    dem=subprocess.run(["c++filt",name],capture_output=True,text=True).stdout.strip()
    if pat not in dem: continue
    g=lambda k: re.search(r"\.%s:\s+(\d+)"%k,blk)
    print(dem[:78],"| vgpr",g("vgpr_count").group(1),"agpr",re.match(r":\s+(\d+)",blk).group(1),"sgpr",g("sgpr_count").group(1),"scratch",g("private_segment_fixed_size").group(1),"lds",g("group_segment_fixed_size").group(1))
