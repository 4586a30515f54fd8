# Ablation builds of conv3x3_dma_kernel (csrc/gemm.hip, macro CD_ABL): 2 = no DMA inside the K loop (MFMA + LDS reads
# only), 3 = no s_barrier per stage.  Results are numerically meaningless; only the timings are.  Run HERE (CPU box):
#   bash tools/exp/conv_ablate.sh          # builds tools/exp/libuwir_abl{2,3}.so from the objects of the last product build
# then on the GPU:  UWIR_EXP_LIB=tools/exp/libuwir_abl2.so python tools/bench_conv.py 50 0 6,10,14,16
# Round-3 result (64^2 C384->256 / 32^2 C768->512): product 58.3 / 61.7 us, no DMA 52.9 / 48.9, no barrier 56.9 / 61.0:
# the MFMA + ds_read loop alone sets ~90 % of the time (1.1-1.2 PFLOP/s at the clock the chip holds under that load).
set -e
cd "$(dirname "$0")/../../underwater_image_restoration_amd"
for v in 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -DCD_ABL=$v -c csrc/gemm.hip -o build/gemm_abl$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/exp/libuwir_abl$v.so $(ls build/*.o | grep -v -e gemm_abl -e 'build/gemm.o' -e dwmerge) build/gemm_abl$v.o
done
