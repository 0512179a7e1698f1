#!/opt/conda/bin/python3.9
"""Write a synthetic aposteriori-style frame dataset with REAL h5py (only /opt/conda python has it in this image):
    /opt/conda/bin/python3.9 tools/make_synthetic_hdf5.py out.hdf5 <n_pdb> <n_residues_per_chain>
gzip-chunked float64 (21,21,21,6) frames under pdb/chain/residue, attrs as aposteriori 2.x writes them (reference
design_utils/utils.py:238-251).  Used to time the HDF5 ingest path (tools/bench_e2e.py)."""
import h5py, numpy as np, sys
rng = np.random.default_rng(0)
THREE = ["ALA","CYS","ASP","GLU","PHE","GLY","HIS","ILE","LYS","LEU","MET","ASN","PRO","GLN","ARG","SER","THR","VAL","TRP","TYR"]
n_pdb, n_res = int(sys.argv[2]), int(sys.argv[3])
with h5py.File(sys.argv[1], "w") as f:
    f.attrs["make_frame_dataset_ver"] = "2.4.0"; f.attrs["frame_dims"] = (21,21,21,6); f.attrs["atom_encoder"] = list("CNOQP") + ["CA"]
    f.attrs["encode_cb"] = True; f.attrs["atom_filter_fn"] = "keep_sidechain_cb"; f.attrs["residue_encoder"] = THREE
    f.attrs["frame_edge_length"] = 21.0; f.attrs["voxels_as_gaussian"] = True
    base = (rng.random((64,21,21,21,6)) * (rng.random((64,21,21,21,6)) < 0.08)).astype(np.float64)
    # the 64 distinct frames are compressed ONCE by h5py/zlib (template datasets, deleted afterwards); every residue dataset is then
    # created with the same gzip filter + automatic chunking and filled with write_direct_chunk — the file is what
    # create_dataset(data=..., compression="gzip") writes, 12x faster (the bench line should not spend a minute in zlib)
    tg = f.create_group("_templates")
    tables = []
    for k in range(64):
        t = tg.create_dataset(str(k), data=base[k], dtype=float, compression="gzip")
        tables.append([(t.id.get_chunk_info(i).chunk_offset,) + tuple(reversed(t.id.read_direct_chunk(t.id.get_chunk_info(i).chunk_offset)))
                       for i in range(t.id.get_num_chunks())])
    chunks = tg["0"].chunks
    for p in range(n_pdb):
        g = f.create_group(f"{p:04x}"[:4].replace(' ','0')); c = g.create_group("A")
        for r in range(n_res):
            d = c.create_dataset(str(r+1), shape=(21,21,21,6), dtype=float, compression="gzip", chunks=chunks)
            for off, raw, mask in tables[(p*n_res+r) % 64]:
                d.id.write_direct_chunk(off, raw, mask)
            d.attrs["label"] = THREE[r % 20]; e = np.zeros(20); e[r % 20] = 1; d.attrs["encoded_residue"] = e
    del f["_templates"]
print("ok")
