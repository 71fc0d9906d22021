for u in 2 4 8 16; do echo "U=$u"; RT_HIP_UNIT_SAMPLES=$u timeout -k 10 120 python3 tools/time_frame.py c2 1920 1080 64 8 10 2>&1 | grep median; done
for u in 4 8 16 32; do echo "C4 128spp U=$u"; RT_HIP_UNIT_SAMPLES=$u timeout -k 10 120 python3 tools/time_frame.py c4 3840 2160 128 8 3 2>&1 | grep median; done
for u in 8 16 32; do echo "C4 1024spp U=$u"; RT_HIP_UNIT_SAMPLES=$u timeout -k 10 120 python3 tools/time_frame.py c4 3840 2160 1024 8 2 2>&1 | grep median; done
