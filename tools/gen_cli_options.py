"""Option names (with click type, default, nargs) of the reference's command-line entry points -> tests/golden/cli_options.json
(build container only: reads /root/reference).  The fixture pins the names this build's entry points must accept."""
import json
import os
import re, sys
def options(path):
    s=open(path).read()
    out={}
    i=0
    while True:
        i=s.find("@click.option(", i)
        if i<0: break
        j=i+len("@click.option("); depth=1
        while depth:
            ch=s[j]
            if ch in "([": depth+=1
            elif ch in ")]": depth-=1
            elif ch in "\"'":
                q=ch; j+=1
                while s[j]!=q: j+= 2 if s[j]=="\\" else 1
            j+=1
        body=s[i+len("@click.option("):j-1]
        names=re.findall(r'^\s*((?:["\']-{1,2}[\w-]+["\']\s*,\s*)+)', body)
        names=re.findall(r'["\'](-{1,2}[\w-]+)["\']', names[0]) if names else []
        long=[n for n in names if n.startswith("--")]
        if long:
            t=re.search(r'type=(click\.\w+(?:\([^)]*\))?)', body)
            d=re.search(r'default=(\([^)]*\)|[^,\n]+)', body)
            n=re.search(r'nargs=(\d+)', body)
            out[long[0]]=dict(names=names, type=t.group(1) if t else None, default=d.group(1).strip() if d else None,
                              nargs=int(n.group(1)) if n else None, required='required=True' in body)
        i=j
    return out
if __name__ == "__main__":
    scripts = ["train_sh_based_voxel_grid_with_posed_images.py", "edit_pretrained_relu_field.py",
               "refine_edited_relu_field.py", "render_sh_based_voxel_grid.py", "render_sh_based_voxel_grid_attn.py",
               "segment_attn_relu_field.py"]
    out = {f: options("/root/reference/" + f) for f in scripts}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cli_options.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path, {f: len(v) for f, v in out.items()})
