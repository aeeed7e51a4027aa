cd /root/repo
timeout 1200 python -m pytest tests/test_roi_pooling_parity.py tests/test_golden_device.py tests/test_hip_parity.py -q -m gpu -k "roi or golden or gather or scatter or pooling or ball or bev_iou_cpu" > gpurun_out/r2_newtests.log 2>&1
tail -n 40 gpurun_out/r2_newtests.log
