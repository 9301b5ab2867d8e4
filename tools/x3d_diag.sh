# gpurun -- bash tools/x3d_diag.sh   (diagnostic library: python -m prodsearch_amd.build --diag)
export PS_DIAG_LIB=1
out=gpurun_out/r04_x3d_diag.txt
: > $out
for e in "PS_X3D_DIAG=0" "PS_X3D_DIAG=1" "PS_X3D_DIAG=2" "PS_X3D_DIAG=3" "PS_X3D_LDSPAD=32768" "PS_X3D_DIAG=1 PS_X3D_LDSPAD=32768" "PS_X3D_DIAG=2 PS_X3D_LDSPAD=32768"; do
  env $e timeout -k 10 120 python tools/x3d_diag.py >> $out 2>&1 || exit 1
done
timeout -k 10 120 python tools/x3d_diag.py 1 >> $out 2>&1
cat $out
