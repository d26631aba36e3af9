"""Edits on disk -- the format either side of the path.

RawEditor persists one JSON row per image in SQLite: table `edits(id, image_id, settings_json)`, newest row
wins (src/state/library.rs:70-78, :310-351); images live in `images(id, path, filename, width, height, ...)`
(:56-66).  These helpers read (and, for tests, write) exactly that schema with the stdlib sqlite3 module so a
batch export can consume the app's own catalog.  `settings_json` is `EditParams::to_json` (edit.rs:105-112).
"""
from __future__ import annotations

import dataclasses
import json
import sqlite3
from typing import List, Optional

from .edit import EditParams


def load_edit_params(conn: sqlite3.Connection, image_id: int) -> EditParams:
    """Library::load_edit_params (library.rs:341-351): newest row; KeyError if the image has no edits
    (rusqlite's QueryReturnedNoRows -- the caller falls back to EditParams::default(), main.rs:510-517)."""
    row = conn.execute("SELECT settings_json FROM edits WHERE image_id = ?1 ORDER BY id DESC LIMIT 1",
                       (int(image_id),)).fetchone()
    if row is None:
        raise KeyError(f"no edits for image {image_id}")
    return EditParams.from_json(row[0])


def load_edit_params_or_default(conn: sqlite3.Connection, image_id: int) -> EditParams:
    try:
        return load_edit_params(conn, image_id)
    except KeyError:
        return EditParams.default()


def save_edit_params(conn: sqlite3.Connection, image_id: int, params: EditParams) -> None:
    """Library::save_edit_params (library.rs:310-337): update the newest row or insert the first."""
    row = conn.execute("SELECT id FROM edits WHERE image_id = ?1 ORDER BY id DESC LIMIT 1", (int(image_id),)).fetchone()
    if row is not None:
        conn.execute("UPDATE edits SET settings_json = ?1 WHERE id = ?2", (params.to_json(), row[0]))
    else:
        conn.execute("INSERT INTO edits (image_id, settings_json) VALUES (?1, ?2)", (int(image_id), params.to_json()))
    conn.commit()


def has_edits(conn: sqlite3.Connection, image_id: int) -> bool:        # library.rs:354-361
    return conn.execute("SELECT COUNT(*) FROM edits WHERE image_id = ?1", (int(image_id),)).fetchone()[0] > 0


def init_schema(conn: sqlite3.Connection) -> None:
    """Library::init_schema (library.rs:52-121), statement for statement, so that a catalog created here opens in the app
    unchanged: the two tables, their indexes, the columns the app adds with ALTER TABLE (the three cache tiers of "Phase 28"
    and file_status; like the app, an "already exists" error from a second run is ignored) and the cache_status index."""
    conn.execute("""CREATE TABLE IF NOT EXISTS images (
                id              INTEGER PRIMARY KEY AUTOINCREMENT,
                path            TEXT NOT NULL UNIQUE,
                filename        TEXT NOT NULL,
                width           INTEGER,
                height          INTEGER,
                imported_at     INTEGER NOT NULL,
                cache_status    TEXT DEFAULT 'pending'
            )""")                                                                       # library.rs:56-66
    conn.execute("""CREATE TABLE IF NOT EXISTS edits (
                id              INTEGER PRIMARY KEY AUTOINCREMENT,
                image_id        INTEGER NOT NULL,
                settings_json   TEXT NOT NULL,
                FOREIGN KEY(image_id) REFERENCES images(id) ON DELETE CASCADE
            )""")                                                                       # library.rs:70-78
    conn.execute("CREATE INDEX IF NOT EXISTS idx_images_imported_at ON images(imported_at DESC)")     # library.rs:81-85
    conn.execute("CREATE INDEX IF NOT EXISTS idx_edits_image_id ON edits(image_id)")                  # library.rs:87-91
    for ddl in ("ALTER TABLE images ADD COLUMN cache_path_thumb TEXT",                                 # library.rs:95-106
                "ALTER TABLE images ADD COLUMN cache_path_instant TEXT",
                "ALTER TABLE images ADD COLUMN cache_path_working TEXT",
                "ALTER TABLE images ADD COLUMN file_status TEXT DEFAULT 'exists'"):                   # library.rs:109-112
        try:
            conn.execute(ddl)
        except sqlite3.OperationalError as e:                    # `let _ = self.conn.execute(...)`: the column is there already
            if "duplicate column" not in str(e):
                raise
    conn.execute("CREATE INDEX IF NOT EXISTS idx_images_cache_status ON images(cache_status)")        # library.rs:115-119
    conn.commit()


@dataclasses.dataclass
class ManifestEntry:
    image_id: int
    path: str
    width: Optional[int]
    height: Optional[int]
    params: EditParams


def export_manifest(conn: sqlite3.Connection, rank: int = 0, world_size: int = 1) -> List[ManifestEntry]:
    """Every image of the catalog with its newest edit (or the default stack), in id order, sharded
    frame i -> rank i mod world_size like raweditor_amd.batch.shard_frames."""
    rows = conn.execute("SELECT id, path, width, height FROM images ORDER BY id").fetchall()
    out = []
    for i, (iid, path, w, h) in enumerate(rows):
        if i % world_size == rank:
            out.append(ManifestEntry(iid, path, w, h, load_edit_params_or_default(conn, iid)))
    return out


def manifest_to_json(entries: List[ManifestEntry]) -> str:
    return json.dumps([{"image_id": e.image_id, "path": e.path, "width": e.width, "height": e.height,
                        "settings_json": e.params.to_json()} for e in entries])


def manifest_from_json(text: str) -> List[ManifestEntry]:
    return [ManifestEntry(d["image_id"], d["path"], d.get("width"), d.get("height"),
                          EditParams.from_json(d["settings_json"])) for d in json.loads(text)]
