for d in 0 1 2 3 4 8 15; do echo "DBG=$d"; CG3D_DBG_WGRAD=$d PREC=1 python tools/microbench_conv.py 2>&1 | grep "128-> 128\|256-> 256\|16->16" | cut -c1-28,262-300; done
