for w in 1 2; do echo "WG_PER_CU=$w"; IMX_DW_WG_PER_CU=$w python tools/gemm_shapes.py | cut -c1-12,100-135; done
