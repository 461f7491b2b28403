"""``scripts/run_detect_contacts.py`` equivalent: OpenPose JSON -> ``foot_contacts.npy`` per video."""
import argparse
import sys

from . import contact_net


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--data', required=True, help='root with one directory per video containing openpose_result/ (run_detect_contacts.py:14)')
    p.add_argument('--weights', required=True, help='op_only_weights.pth state_dict (pretrained_weights/download.sh)')
    p.add_argument('--full-video', action='store_true'); p.add_argument('--save-contacts', action='store_true'); p.add_argument('--real-data', action='store_true')
    p.add_argument('--width', type=int, default=1920); p.add_argument('--height', type=int, default=1080)
    p.add_argument('--device-ops', action='store_true', help='(the default since round 5; accepted for old command lines)')
    p.add_argument('--host-ops', action='store_true', help='gap interpolation, windowing and vote merge in NumPy on the host instead of tensor ops on the device (same labels, ~17 x slower end to end)')
    a = p.parse_args(sys.argv[1:] if argv is None else argv)
    res = contact_net.run_on_directory(a.data, a.weights, dimensions=(a.width, a.height), device_ops=not a.host_ops)
    print('[run_detect_contacts] wrote foot_contacts.npy for %d videos on %s' % (len(res), contact_net.select_device()))
    return 0


if __name__ == '__main__':
    sys.exit(main())
