from ripor_amd.evaluate import *  # noqa: F401,F403
from ripor_amd.evaluate import main

if __name__ == "__main__":
    main()
