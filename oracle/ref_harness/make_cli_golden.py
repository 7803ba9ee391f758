"""Golden data for the command-line drop-in (build container only; test infrastructure).

    python -m oracle.ref_harness.make_cli_golden     # writes tests/golden/cli_reference_args.json, tests/golden/process_list_autogen.csv

1. `cli_reference_args.json`: every `add_argument` call of the reference's four entry points on the path, read off their source with
   `ast` (nothing is imported or executed): flags, dest, default, type name, action, required, nargs.
       tools/infer.py:17-38, tools/infer_wsi.py:309-356, tools/infer_patch.py:106-190, tools/nuclei_merge.py:221-230
2. `process_list_autogen.csv`: the text `seg_and_patch` writes (`df.to_csv(..., index=False)`, tools/infer_wsi.py:159,291) for three
   slide names, produced by the REFERENCE's own `initialize_df` (tools/wsi_core/batch_process_utils.py:17-82, imported as a file;
   needs only numpy + pandas) with the parameter dicts of tools/infer_wsi.py:378-382.
"""
import ast
import importlib.util
import json
import os

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden')
TOOLS = ['tools/infer.py', 'tools/infer_wsi.py', 'tools/infer_patch.py', 'tools/nuclei_merge.py']


def _lit(node):
    if isinstance(node, ast.Name):
        return node.id                       # type=int -> "int"
    try:
        return ast.literal_eval(node)
    except Exception:
        return ast.unparse(node)


def parser_table(path):
    tree = ast.parse(open(path).read())
    rows = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'add_argument':
            flags = [_lit(a) for a in node.args]
            kw = {k.arg: _lit(k.value) for k in node.keywords if k.arg != 'help'}
            rows.append(dict(flags=flags, line=node.lineno, **kw))
    rows.sort(key=lambda r: r['line'])
    return rows


def main():
    os.makedirs(OUT, exist_ok=True)
    table = {t: parser_table(os.path.join(REF, t)) for t in TOOLS}
    with open(os.path.join(OUT, 'cli_reference_args.json'), 'w') as f:
        json.dump(table, f, indent=1, sort_keys=True)
    spec = importlib.util.spec_from_file_location('ref_batch_process_utils', os.path.join(REF, 'tools/wsi_core/batch_process_utils.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seg_params = {'seg_level': -1, 'sthresh': 8, 'mthresh': 7, 'close': 4, 'use_otsu': False, 'keep_ids': 'none', 'exclude_ids': 'none'}
    filter_params = {'a_t': 100, 'a_h': 16, 'max_n_holes': 8}
    vis_params = {'vis_level': -1, 'line_thickness': 250}
    patch_params = {'use_padding': True, 'contour_fn': 'four_pt'}
    df = mod.initialize_df(['a.npy', 'b.npy', 'c.svs'], seg_params, filter_params, vis_params, patch_params)
    # what the loop does to a processed slide (:163,253-254,289) and to an auto-skipped one (:169-171)
    df.loc[0, 'process'] = 0
    df.loc[0, 'vis_level'] = 6
    df.loc[0, 'seg_level'] = 6
    df.loc[0, 'status'] = 'processed'
    df.loc[1, 'process'] = 0
    df.loc[1, 'status'] = 'already_exist'
    df.to_csv(os.path.join(OUT, 'process_list_autogen.csv'), index=False)
    print(open(os.path.join(OUT, 'process_list_autogen.csv')).read())


if __name__ == '__main__':
    main()
