// placeholder: filled in below
