#!/usr/bin/env python3
"""``python train_sr.py ...`` -- same command line as the reference's train_sr.py; see amid_amd/train_sr.py."""
from amid_amd.train_sr import main

if __name__ == "__main__":
    main()
