from ripor_amd.main import *  # noqa: F401,F403
from ripor_amd.main import main

if __name__ == "__main__":
    main()
