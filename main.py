"""Drop-in entry point: `python main.py -p train --config_json configs/config.json --gpu 0`
(same flags as the reference's main.py:22-48)."""
from vnet_tensorflow_amd.main import get_parser, main

if __name__ == "__main__":
    main(get_parser())
