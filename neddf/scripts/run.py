import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from neddf_amd.scripts.run import main, seed_everything  # noqa: E402

if __name__ == "__main__":
    seed_everything()
    main()
