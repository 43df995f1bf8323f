"""Build container only: read every YAML entry point of the reference (configs/**/*.yaml) through
mvgformer_amd.factory.load_yaml_config and record the hot-path hyper-parameters it yields as
mvgformer_amd/data/yaml_extract.json -- values only (no reference text), so that the GPU box, which has no /root/reference,
can still build the decoders these files describe.  tests/test_abi.py opens the real files when they exist and
compares them with this extract."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("MVG_REFERENCE", "/root/reference")


def flatten(cfg):
    return {"DECODER": dict(vars(cfg.DECODER)), "IMAGE_SIZE": cfg.NETWORK.IMAGE_SIZE,
            "SPACE_SIZE": cfg.MULTI_PERSON.SPACE_SIZE, "SPACE_CENTER": cfg.MULTI_PERSON.SPACE_CENTER,
            "CAMERA_NUM": cfg.DATASET.CAMERA_NUM}


def main():
    from mvgformer_amd.factory import load_yaml_config
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "configs", "**", "*.yaml"), recursive=True)):
        out[os.path.relpath(path, REF)] = flatten(load_yaml_config(path))
    with open(os.path.join(ROOT, "mvgformer_amd", "data", "yaml_extract.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("%d YAML entry points" % len(out))


if __name__ == "__main__":
    main()
