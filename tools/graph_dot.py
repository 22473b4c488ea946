"""Node -> stream table of a graph printed by DEBUG_HIP_GRAPH_DOT_PRINT=1 (tools/r05_graph_env.sh):  python tools/graph_dot.py <file>"""
import re,sys
s=open(sys.argv[1]).read()
nodes={}
for m in re.finditer(r'"graph_1_node_(\d+)"\[[^\]]*label="(\d+)\n([^\n]*)\nStreamId:(\d+)\nSignalIsRequired: (\w+)',s):
    i=int(m.group(1)); name=m.group(3)
    mm=re.search(r'(sg_[a-z0-9_]+|at6native\d+[a-z_]+)',name)
    nodes[i]=(mm.group(1)[:28] if mm else name[:28], int(m.group(4)), m.group(5))
edges=[(int(a),int(b)) for a,b in re.findall(r'"graph_1_node_(\d+)" -> "graph_1_node_(\d+)"',s)]
preds={}; succ={}
for a,b in edges: preds.setdefault(b,[]).append(a); succ.setdefault(a,[]).append(b)
for i in sorted(nodes):
    n=nodes[i]
    print(f"{i:3d} s{n[1]} {'SIG' if n[2]=='true' else '   '} {n[0]:28s} <- {[f'{a}(s{nodes[a][1]})' for a in preds.get(i,[]) if a in nodes]}  -> {succ.get(i,[])}")
