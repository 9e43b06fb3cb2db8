mkdir -p gpurun_out
timeout 900 python tools/e2e_ab.py 15 METHEOR_COPY_PIECE_MB=0 METHEOR_COPY_PIECE_MB=256 METHEOR_FIRST_CHUNK_MB=1024,METHEOR_COPY_PIECE_MB=256 METHEOR_COPY_PIECE_MB=170 > gpurun_out/e2e_ab_piece2.log 2>&1
cat gpurun_out/e2e_ab_piece2.log
