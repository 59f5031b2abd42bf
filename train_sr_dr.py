#!/usr/bin/env python3
"""``python train_sr_dr.py ...`` -- same command line as the reference's train_sr_dr.py (run.sh); see amid_amd/train_sr_dr.py."""
from amid_amd.train_sr_dr import main

if __name__ == "__main__":
    main()
