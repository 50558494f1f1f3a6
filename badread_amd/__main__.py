"""
`badread simulate` command line for the MI355X path (python -m badread_amd simulate ...).

Flags, defaults, derived fields, validation messages and exit codes follow the reference CLI
(/root/reference/badread/__main__.py:83-147 flags, :239-336 checks) so that existing command lines and
scripts keep working.  `error_model` and `qscore_model` (the model builders) run their counting
loops on the GPU (badread_amd/model_builder.py); `plot` is outside this build's scope and refused with a message.  Additive options:
--gpu-batch (reads per device batch).  Multi-GPU: launch with
`python -m torch.distributed.run --nproc-per-node N -m badread_amd simulate ...`; rank 0 writes stdout.
"""
import argparse
import os
import pathlib
import sys

# A hardware queue per HIP stream of every batch in flight (8 batches x up to 3 streams): HIP's default of 4 queues
# serialises kernels of different streams.  Must be in the environment before the HIP runtime starts (the first torch.cuda
# call), which is why it is set here, at the top of the command line's module, and not where the streams are created.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '40')

from . import settings
from .misc import str_is_dna_sequence, str_is_int
from .version import __version__

ERROR_MODEL_NAMES = ['random', 'nanopore2018', 'nanopore2020', 'nanopore2023', 'pacbio2016', 'pacbio2021']
QSCORE_MODEL_NAMES = ['random', 'ideal', 'nanopore2018', 'nanopore2020', 'nanopore2023', 'pacbio2016', 'pacbio2021']
OUT_OF_SCOPE = ('plot',)

# (group title, group description, [(flag, kwargs), ...]) -- help texts as in the reference
SIMULATE_OPTIONS = [
    ('Required arguments', None, [
        ('--reference', dict(type=str, required=True, help='Reference FASTA file (can be gzipped)')),
        ('--quantity', dict(type=str, required=True,
                            help='Either an absolute value (e.g. 250M) or a relative depth (e.g. 25x)')),
    ]),
    ('Simulation parameters', 'Length and identity and error distributions', [
        ('--length', dict(type=str, default='15000,13000',
                          help='Fragment length distribution (mean and stdev, default: %(default)s)')),
        ('--identity', dict(type=str, default='95,99,2.5',
                            help='Sequencing identity distribution (mean,max,stdev for beta distribution or '
                                 'mean,stdev for normal qscore distribution, default: %(default)s)')),
        ('--error_model', dict(type=str, default='nanopore2023',
                               help='Can be "nanopore2018", "nanopore2020", "nanopore2023", "pacbio2016", '
                                    '"pacbio2021", "random" or a model filename')),
        ('--qscore_model', dict(type=str, default='nanopore2023',
                                help='Can be "nanopore2018", "nanopore2020", "nanopore2023", "pacbio2016", '
                                     '"pacbio2021", "random", "ideal" or a model filename')),
        ('--seed', dict(type=int, help='Random number generator seed for deterministic output (default: '
                                       'different output each time)')),
    ]),
    ('Adapters', 'Controls adapter sequences on the start and end of reads', [
        ('--start_adapter', dict(type=str, default='90,60',
                                 help='Adapter parameters for read starts (rate and amount, default: %(default)s)')),
        ('--end_adapter', dict(type=str, default='50,20',
                               help='Adapter parameters for read ends (rate and amount, default: %(default)s)')),
        ('--start_adapter_seq', dict(type=str, default='AATGTACTTCGTTCAGTTACGTATTGCT',
                                     help='Adapter sequence for read starts')),
        ('--end_adapter_seq', dict(type=str, default='GCAATACGTAACTGAACGAAGT', help='Adapter sequence for read ends')),
    ]),
    ('Problems', 'Ways reads can go wrong', [
        ('--junk_reads', dict(type=float, default=1, help='This percentage of reads will be low-complexity junk')),
        ('--random_reads', dict(type=float, default=1, help='This percentage of reads will be random sequence')),
        ('--chimeras', dict(type=float, default=1, help='Percentage at which separate fragments join together')),
        ('--glitches', dict(type=str, default='10000,25,25',
                            help='Read glitch parameters (rate, size and skip, default: %(default)s)')),
        ('--small_plasmid_bias', dict(action='store_true',
                                      help='If set, then small circular plasmids are lost when the fragment length is '
                                           'too high (default: small plasmids are included regardless of fragment '
                                           'length)')),
    ]),
    ('MI355X', 'Device options (additive; no effect on the simulated reads)', [
        ('--gpu-batch', dict(type=int, default=None, dest='gpu_batch',
                             help='Maximum reads per device batch and GPU (default: 65536)')),
        ('--gzip', dict(type=int, default=None, dest='gzip_level', metavar='LEVEL',
                        help='Write gzip-compressed FASTQ to stdout, compressed on all host cores (level 0-9); '
                             'default: plain text, as the reference')),
        ('--gzip-device', dict(action='store_true', dest='gzip_device',
                               help='Write gzip-compressed FASTQ to stdout, compressed on the GPU before the bytes cross PCIe '
                                    '(a Huffman code per sequence line and per quality line, no match search: smaller than '
                                    'gzip -6 on simulated reads, and the host only copies)')),
        ('--output-shards', dict(type=str, default=None, dest='output_shards', metavar='PREFIX',
                                 help='Multi-GPU runs: every rank writes the records of ITS reads to PREFIX.<rank>.fastq (.fastq.gz with '
                                      '--gzip / --gzip-device) and the bytes per batch to PREFIX.<rank>.parts (FASTQ text bytes; with '
                                      '--gzip-device the compressed bytes; with --gzip the compressed bytes per batch go to '
                                      'PREFIX.<rank>.zparts beside them), instead of all records travelling to rank 0 and through one '
                                      'stdout; the same reads, the same stopping point')),
        ('--gpu-streams', dict(type=int, default=None, dest='gpu_streams',
                               help='Device batches in flight per GPU, each on its own HIP stream (default: 6)')),
    ]),
]


class _Parser(argparse.ArgumentParser):
    """Usage errors print the help and exit with status 2, like the reference's parser."""

    def error(self, message):
        self.print_help(file=sys.stderr)
        sys.exit(2)


def _add_help_group(parser):
    other = parser.add_argument_group('Other' if parser.prog.endswith('simulate') else 'Help')
    other.add_argument('-h', '--help', action='help', default=argparse.SUPPRESS, help='Show this help message and exit')
    other.add_argument('--version', action='version', version='Badread v' + __version__,
                       help="Show program's version number and exit")


def parse_args(argv):
    parser = _Parser(prog='badread', add_help=False,
                     description='Badread: a long read simulator that can imitate many types of read problems '
                                 '(MI355X-native simulate path)')
    subparsers = parser.add_subparsers(title='Commands', dest='subparser_name')
    sim = subparsers.add_parser('simulate', description='Generate fake long reads', add_help=False)
    for title, description, options in SIMULATE_OPTIONS:
        group = sim.add_argument_group(title, description=description)
        for flag, kwargs in options:
            group.add_argument(flag, **kwargs)
    _add_help_group(sim)
    # the model builders: flags and defaults of the reference (__main__.py:150-205)
    em = subparsers.add_parser('error_model', description='Build a Badread error model', add_help=False)
    qm = subparsers.add_parser('qscore_model', description='Build a Badread qscore model', add_help=False)
    for sub in (em, qm):
        req = sub.add_argument_group('Required arguments')
        req.add_argument('--reference', type=str, required=True, help='Reference FASTA file')
        req.add_argument('--reads', type=str, required=True, help='FASTQ of real reads')
        req.add_argument('--alignment', type=str, required=True, help='PAF alignment of reads aligned to reference')
    opt = em.add_argument_group('Optional arguments')
    opt.add_argument('--k_size', type=int, default=7, help='Error model k-mer size')
    opt.add_argument('--max_alignments', type=int, help='Only use this many alignments when generating error model '
                                                        '(default: use all alignments)')
    opt.add_argument('--max_alt', type=int, default=25, help='Only save up to this many alternatives to each k-mer')
    opt = qm.add_argument_group('Optional arguments')
    opt.add_argument('--k_size', type=int, default=9, help='Qscore model k-mer size (must be odd, default: %(default)s)')
    opt.add_argument('--max_alignments', type=int, help='Only use this many alignments when generating qscore model '
                                                        '(default: use all alignments)')
    opt.add_argument('--max_del', type=int, default=6, help='Deletion runs longer than this will be collapsed to reduce '
                                                            'the number of possible alignments')
    opt.add_argument('--min_occur', type=int, default=100, help='CIGARs which occur less than this many times will not be '
                                                                'included in the model')
    opt.add_argument('--max_output', type=int, default=10000, help='The outputted model will be limited to this many lines')
    _add_help_group(em)
    _add_help_group(qm)
    for name in OUT_OF_SCOPE:
        sub = subparsers.add_parser(name, add_help=False, description=f'{name}: not part of the MI355X build')
        sub.add_argument('rest', nargs=argparse.REMAINDER)
    _add_help_group(parser)
    if len(argv) == 0:
        parser.print_help(file=sys.stderr)
        sys.exit(1)
    return parser.parse_args(argv)


def _floats(text, count=None):
    values = [float(x) for x in text.split(',')]
    if count is not None and len(values) < count:
        raise IndexError(text)
    return values


def check_simulate_args(args):
    """Validate and derive the fields simulate() reads (mean_frag_length, identity triple, glitch_*)."""
    if not pathlib.Path(args.reference).is_file():
        sys.exit(f'Error: {args.reference} is not a file')
    for value, names, flag in ((args.error_model, ERROR_MODEL_NAMES, '--error_model'),
                               (args.qscore_model, QSCORE_MODEL_NAMES, '--qscore_model')):
        if value.lower() not in names and not pathlib.Path(value).is_file():
            sys.exit(f'Error: {value} is not a file\n  {flag} must be from {names} or a filename')

    for limit_hit, message in ((args.chimeras > 50, '--chimeras cannot be greater than 50'),
                               (args.junk_reads > 100, '--junk_reads cannot be greater than 100'),
                               (args.random_reads > 100, '--random_reads cannot be greater than 100'),
                               (args.junk_reads + args.random_reads > 100,
                                '--junk_reads and --random_reads cannot sum to more than 100')):
        if limit_hit:
            sys.exit('Error: ' + message)

    try:
        args.mean_frag_length, args.frag_length_stdev = _floats(args.length, 2)[:2]
    except (ValueError, IndexError):
        sys.exit('Error: could not parse --length values')
    if args.mean_frag_length <= settings.MIN_MEAN_READ_LENGTH:
        sys.exit(f'Error: mean read length must be at least {settings.MIN_MEAN_READ_LENGTH}')
    if args.frag_length_stdev < 0:
        sys.exit('Error: read length stdev cannot be negative')

    try:
        identity = _floats(args.identity)
    except ValueError:
        sys.exit('Error: could not parse --identity values')
    if len(identity) == 2:
        args.mean_identity, args.max_identity, args.identity_stdev = identity[0], None, identity[1]
        check_qscore_identities(args)
    elif len(identity) == 3:
        args.mean_identity, args.max_identity, args.identity_stdev = identity
        check_beta_identities(args)
    else:
        sys.exit('Error: could not parse --identity values')

    try:
        args.glitch_rate, args.glitch_size, args.glitch_skip = _floats(args.glitches, 3)[:3]
    except (ValueError, IndexError):
        sys.exit('Error: could not parse --glitches values')
    if min(args.glitch_rate, args.glitch_size, args.glitch_skip) < 0:
        sys.exit('Error: --glitches must contain non-negative values')

    for attr, flag in (('start_adapter_seq', '--start_adapter_seq'), ('end_adapter_seq', '--end_adapter_seq')):
        value = getattr(args, attr)
        if value != '' and not str_is_int(value):
            value = value.upper()
            setattr(args, attr, value)
            if not str_is_dna_sequence(value):
                sys.exit(f'Error: {flag} must be a DNA sequence or a number')


def check_beta_identities(args):
    floor = settings.MIN_MEAN_READ_IDENTITY
    checks = ((args.mean_identity > 100.0, 'mean read identity cannot be more than 100'),
              (args.max_identity > 100.0, 'max read identity cannot be more than 100'),
              (args.mean_identity <= floor, f'mean read identity must be at least {floor}'),
              (args.max_identity <= floor, f'max read identity must be at least {floor}'),
              (args.mean_identity > args.max_identity,
               f'mean identity ({args.mean_identity}) cannot be larger than max identity ({args.max_identity})'),
              (args.identity_stdev < 0.0, 'read identity stdev cannot be negative'))
    for failed, message in checks:
        if failed:
            sys.exit('Error: ' + message)


def check_qscore_identities(args):
    if args.mean_identity <= settings.MIN_MEAN_READ_QSCORE:
        sys.exit(f'Error: mean read identity must be at least {settings.MIN_MEAN_READ_QSCORE}')
    if args.identity_stdev < 0.0:
        sys.exit('Error: read qscore stdev cannot be negative')


def check_python_version():
    if sys.version_info.major < 3 or sys.version_info.minor < 6:
        sys.exit('Error: Badread requires Python 3.6 or later')


def main(output=sys.stderr):
    check_python_version()
    args = parse_args(sys.argv[1:])
    if args.subparser_name == 'simulate':
        check_simulate_args(args)
        from . import simulate as sim
        if args.gpu_batch:
            sim.DEFAULT_MAX_BATCH = int(args.gpu_batch)
        sim.simulate(args, output=output)
    elif args.subparser_name == 'error_model':
        from .model_builder import make_error_model
        make_error_model(args, output=output)
    elif args.subparser_name == 'qscore_model':
        from .model_builder import make_qscore_model
        make_qscore_model(args, output=output)
    elif args.subparser_name in OUT_OF_SCOPE:
        sys.exit(f'Error: the {args.subparser_name} command is not part of the MI355X simulate build; '
                 f'use the reference Badread for it')
    else:
        parse_args([])


if __name__ == '__main__':
    main()
